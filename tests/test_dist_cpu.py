"""CPU, world_size 2, gloo: the N>1 path -- one flat-bucket gradient all-reduce over the replicated Q-network
(SURVEY.md section 8e), shard ownership of envs/graphs, and edge-balanced partition of ragged batches."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import model_args


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gnn_hex_amd.dist import GradSync
    from gnn_hex_amd.models import get_pre_defined
    torch.manual_seed(0)                                  # identical replicas
    model = get_pre_defined("modern_two_headed", model_args(3, 8))
    sync = GradSync(model.parameters())
    # every rank works on the maker side this step: breaker-head parameters keep grad None on all ranks
    gen = torch.Generator().manual_seed(100 + rank)
    expect = {}
    for k, p in model.named_parameters():
        if k.startswith("breaker_head"):
            continue
        p.grad = torch.randn(p.shape, generator=gen)
    # reference result: average of the two ranks' gradients, recomputed locally from both seeds
    gens = [torch.Generator().manual_seed(100 + r) for r in range(world)]
    for k, p in model.named_parameters():
        if k.startswith("breaker_head"):
            continue
        expect[k] = sum(torch.randn(p.shape, generator=g) for g in gens) / world
    nelem = sync.all_reduce()
    ok = nelem == sum(p.numel() for k, p in model.named_parameters() if not k.startswith("breaker_head"))
    for k, p in model.named_parameters():
        if k.startswith("breaker_head"):
            ok = ok and p.grad is None
        else:
            ok = ok and torch.allclose(p.grad, expect[k], atol=1e-6)
    # second step with the other side: the bucket is rebuilt for the new participating set
    for p in model.parameters():
        p.grad = None
    for k, p in model.named_parameters():
        if not k.startswith("maker_head"):
            p.grad = torch.full(p.shape, float(rank + 1))
    sync.all_reduce()
    for k, p in model.named_parameters():
        if k.startswith("maker_head"):
            ok = ok and p.grad is None
        else:
            ok = ok and torch.allclose(p.grad, torch.full(p.shape, 1.5))
    # third step: gradients that already live in ONE flat buffer (what the fused backward produces) are reduced in
    # place, without the copy in / copy out
    for p in model.parameters():
        p.grad = None
    active = [p for k, p in model.named_parameters() if not k.startswith("breaker_head")]
    flat = torch.full((sum(p.numel() for p in active),), float(rank + 1))
    views = torch._C._nn.unflatten_dense_tensors(flat, active)
    for p, v in zip(active, views):
        p.grad = v
    ok = ok and sync._adopt_flat(active) is not None
    sync.all_reduce()
    ok = ok and torch.allclose(flat, torch.full_like(flat, 1.5))
    ok = ok and all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(active, views))
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_flat_bucket_all_reduce_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=100) for _ in procs]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def test_shard_range_and_edge_balance():
    from gnn_hex_amd.dist import balance_by_edges, shard_range
    assert [shard_range(1024, r, 8) for r in range(8)] == [(128 * r, 128 * (r + 1)) for r in range(8)]
    cover = []
    for r in range(3):
        lo, hi = shard_range(10, r, 3)
        cover += list(range(lo, hi))
    assert cover == list(range(10))
    edges = [116, 174, 244, 326, 420, 526, 644, 774, 916] * 4      # Hex-5..13 round robin
    parts = balance_by_edges(edges, 8)
    assert sorted(i for p in parts for i in p) == list(range(36))
    loads = [sum(edges[i] for i in p) for p in parts]
    assert max(loads) - min(loads) <= max(edges)
    assert parts == balance_by_edges(edges, 8)                       # deterministic
