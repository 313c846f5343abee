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
    # fourth step: the ranks disagree on the side (rank 0 maker head, rank 1 breaker head: equal element counts, so the
    # plain bucket would silently mix the two heads) -- check=True must raise on every rank instead of reducing
    checked = GradSync(model.parameters(), check=True)
    for p in model.parameters():
        p.grad = None
    skip = "breaker_head" if rank == 0 else "maker_head"
    for k, p in model.named_parameters():
        if not k.startswith(skip):
            p.grad = torch.ones(p.shape)
    try:
        checked.all_reduce()
        ok = False
    except RuntimeError as exc:
        ok = ok and "different parameter sets" in str(exc)
    # ... and passes when they agree
    for p in model.parameters():
        p.grad = None
    for k, p in model.named_parameters():
        if not k.startswith("breaker_head"):
            p.grad = torch.full(p.shape, float(rank))
    checked.all_reduce()
    ok = ok and all(torch.allclose(p.grad, torch.full(p.shape, 0.5)) for p in model.parameters() if p.grad is not None)
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


def _worker8(rank, world, port, q):
    """The N = 8 leg (BASELINE config 4: 1024 envs sharded 8 ways) on CPU: one flat bucket through gloo with the same-set
    check on, a strong-scaling partition of a ragged 256-graph batch known identically to every rank, and a rank that
    trains the other side caught by the check."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gnn_hex_amd.dist import GradSync, balance_by_edges, shard_range
    from gnn_hex_amd.models import get_pre_defined
    torch.manual_seed(0)
    model = get_pre_defined("modern_two_headed", model_args(3, 8))
    sync = GradSync(model.parameters(), check=True)
    active = [p for k, p in model.named_parameters() if not k.startswith("breaker_head")]
    flat = torch.full((sum(p.numel() for p in active),), float(rank + 1))
    for p, v in zip(active, torch._C._nn.unflatten_dense_tensors(flat, active)):
        p.grad = v
    n = sync.all_reduce()
    ok = n == flat.numel() and torch.allclose(flat, torch.full_like(flat, (world + 1) / 2.0))
    # strong scaling of ONE ragged batch: every rank computes the same partition and owns a disjoint part; loss weights
    # k_r * world / B make the average of the per-rank means the global mean
    sizes = [5 + (g % 9) for g in range(256)]
    edges = [2 * (2 * s + (s - 2) * (s - 1) + s * (s - 1) + (s - 1) ** 2) for s in sizes]
    parts = balance_by_edges(edges, world)
    mine = parts[rank]
    cnt = torch.tensor([float(len(mine)), float(sum(edges[i] for i in mine)), float(sum(mine))])
    gathered = [torch.zeros(3) for _ in range(world)]
    dist.all_gather(gathered, cnt)
    tot = torch.stack(gathered).sum(0)
    ok = ok and int(tot[0]) == 256 and int(tot[1]) == sum(edges) and int(tot[2]) == sum(range(256))
    loads = torch.stack(gathered)[:, 1]
    ok = ok and float(loads.max() - loads.min()) <= max(edges)
    w = len(mine) * world / 256.0                    # this rank's loss weight
    contrib = torch.tensor([w * 1.0])                # a per-rank mean of 1 -> global mean must be 1
    dist.all_reduce(contrib)
    ok = ok and abs(float(contrib) / world - 1.0) < 1e-6
    ok = ok and shard_range(1024, rank, world) == (128 * rank, 128 * (rank + 1))
    # rank 5 trains the other side: every rank must raise instead of reducing
    for p in model.parameters():
        p.grad = None
    skip = "maker_head" if rank == 5 else "breaker_head"
    for k, p in model.named_parameters():
        if not k.startswith(skip):
            p.grad = torch.ones(p.shape)
    try:
        sync.all_reduce()
        ok = False
    except RuntimeError as exc:
        ok = ok and "different parameter sets" in str(exc)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_eight_rank_gloo_rehearsal():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(r, True) for r in range(8)]


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


_CHILD = """
import os, sys
import torch, torch.distributed as dist
dist.init_process_group("gloo")
t = torch.tensor([float(dist.get_rank() + 1)])
dist.all_reduce(t)
if os.environ.get("FAIL_RANK") == os.environ["RANK"]:
    sys.exit(7)
if dist.get_rank() == 0:
    print("sum=%d world=%s local=%s" % (int(t.item()), os.environ["WORLD_SIZE"], os.environ["LOCAL_RANK"]))
dist.barrier()
"""


@pytest.mark.timeout(180)
def test_launch_ranks_brings_up_its_own_world(tmp_path, capfd):
    """gnn_hex_amd.dist.launch_ranks is what `bench.py --gpus N` uses when no launcher set WORLD_SIZE: N fresh child
    processes with the rendezvous environment, rank 0's stdout passed through, a failing rank's code returned."""
    import sys
    from gnn_hex_amd.dist import launch_ranks
    script = tmp_path / "child.py"
    script.write_text(_CHILD)
    assert launch_ranks([sys.executable, str(script)], 3, timeout_s=150) == 0
    assert "sum=6 world=3 local=0" in capfd.readouterr().out
    os.environ["FAIL_RANK"] = "1"
    try:
        assert launch_ranks([sys.executable, str(script)], 2, timeout_s=150) == 7
    finally:
        del os.environ["FAIL_RANK"]


def test_bench_refuses_mismatched_world(monkeypatch):
    """bench.py must never print a line whose n_gpus differs from --gpus: under a launcher with the wrong WORLD_SIZE it
    exits, and without GPUs for the requested ranks it says so instead of silently running one rank."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=2 but --gpus=4" in r.stderr
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    import torch
    if torch.cuda.device_count() < 8:
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8"], env=env, capture_output=True,
                           text=True, timeout=120)
        assert r.returncode != 0 and "only" in r.stderr and "GPU(s) visible" in r.stderr and r.stdout.strip() == ""
