"""Staged backward for the gradient all-reduce overlap: with a stage hook installed the fused backward issues
[data chain + small reduces + weight gradients of the upper hidden layers], hands the TAIL of the flat gradient buffer to
the hook, then issues the lower layers' weight-gradient GEMM and hands over the head of the buffer.  The two segments
tile the buffer exactly, the first one is handed over BEFORE the second GEMM is enqueued, and the gradients equal the
single-call backward (and the oracle)."""
import pytest
import torch

from helpers import batch_tensors, make_pair, sel_and_targets

pytestmark = pytest.mark.gpu


def _step(model, data):
    model.zero_grad(set_to_none=True)
    x, ei, batch, ptr, sel, tgt = data
    q = model(x, ei, batch, ptr)
    torch.nn.functional.mse_loss(q[sel], tgt).backward()
    return q.detach(), {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in model.named_parameters()}


@pytest.mark.parametrize("math", ["fp32", "f16x3"])
@pytest.mark.parametrize("layers,hidden,sizes", [(15, 110, [11] * 40), (10, 35, [7] * 64), (3, 16, [5, 7])])
def test_staged_backward_matches_single_call(math, layers, hidden, sizes):
    from gnn_hex_amd import ops
    from gnn_hex_amd.dist import GradSync
    ops.set_math(math)
    try:
        hip, ref = make_pair(layers, hidden, seed=61)
        x, ei, batch, ptr = batch_tensors("D1", sizes, maker=False)
        sel, tgt = sel_and_targets(ptr)
        data = [t.cuda() for t in (x, ei, batch, ptr, sel, tgt)]
        q0, g0 = _step(hip, data)
        calls = []

        def hook(flat, lo, hi):
            ev = torch.cuda.Event()
            ev.record()
            calls.append((flat.data_ptr(), flat.numel(), lo, hi, ev))

        ops.set_grad_stage_hook(hook)
        q1, g1 = _step(hip, data)
        torch.cuda.synchronize()
        # two segments, tail first, tiling the flat buffer of this step's gradients
        assert len(calls) == 2
        (p0, n0, lo0, hi0, _), (p1, n1, lo1, hi1, _) = calls
        assert p0 == p1 and n0 == n1 and hi0 == n0 and lo1 == 0 and hi1 == lo0 and 0 < lo0 < n0
        active = [p for p in hip.parameters() if p.grad is not None]
        flat = GradSync._adopt_flat(active)
        assert flat is not None and flat.numel() == n0 and flat.data_ptr() == p0
        # the boundary is a layer boundary: the head segment holds the first 1 + tot // 2 conv layers
        tot = layers + 2
        mid = 1 + tot // 2
        convs = list(hip.gnn.convs) + list(hip.breaker_head.gnn.convs)
        head_elems = sum(c.lin_l.weight.numel() + c.lin_l.bias.numel() + c.lin_r.weight.numel() for c in convs[:mid])
        assert lo0 == head_elems
        assert torch.equal(q0, q1)
        for k in g0:
            assert (g0[k] is None) == (g1[k] is None), k
            if g0[k] is not None:
                scale = max(1.0, g0[k].abs().max().item())
                assert (g0[k] - g1[k]).abs().max().item() < 2e-6 * scale, k
        # and against the oracle at the usual bar
        ref.zero_grad(set_to_none=True)
        q_ref = ref(x, ei, batch, ptr)
        torch.nn.functional.mse_loss(q_ref[sel], tgt).backward()
        for k, p in ref.named_parameters():
            if p.grad is not None:
                assert (g1[k].cpu() - p.grad).abs().max().item() < 1e-4 * max(1.0, p.grad.abs().max().item()), k
    finally:
        ops.set_grad_stage_hook(None)
        ops.set_math("fp32")


def test_first_segment_is_handed_over_before_the_second_gemm_finishes():
    """Enqueue-order evidence on one GPU: an event recorded inside the first hook call completes strictly before the
    step's last kernel (the second stage's reduce), with the second weight-gradient GEMM in between."""
    from gnn_hex_amd import ops
    hip, _ = make_pair(15, 110, seed=62)
    x, ei, batch, ptr = batch_tensors("D0", [11] * 256)
    sel, tgt = sel_and_targets(ptr)
    data = [t.cuda() for t in (x, ei, batch, ptr, sel, tgt)]
    events = []

    def hook(flat, lo, hi):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        events.append(ev)

    try:
        for _ in range(3):
            _step(hip, data)
        ops.set_grad_stage_hook(hook)
        for _ in range(3):
            events.clear()
            _step(hip, data)
            torch.cuda.synchronize()
        gap_ms = events[0].elapsed_time(events[1])
        print("first segment final %.1f us before the second: the window an all-reduce overlaps with" % (gap_ms * 1e3))
        assert gap_ms * 1e3 > 40.0          # the lower half of the weight-gradient GEMM (~100 us at GNN-L B=256)
    finally:
        ops.set_grad_stage_hook(None)


def test_rccl_call_sequence_on_a_one_rank_group():
    """The N-rank path's exact RCCL calls on the one GPU this box has: a 1-rank "nccl" process group (device bound), the
    staged backward handing two segments of the flat gradient buffer to async ReduceOp.AVG all-reduces on the group's
    stream, all_reduce() waiting for them, the same-set check, and the plain one-bucket path -- gradients must come out
    unchanged (an average over one rank), bit for bit."""
    import os
    import socket
    import torch.distributed as dist
    from gnn_hex_amd import ops
    from gnn_hex_amd.dist import GradSync
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        hip, _ = make_pair(10, 35, seed=63)
        x, ei, batch, ptr = batch_tensors("D1", [7] * 64)
        sel, tgt = sel_and_targets(ptr)
        data = [t.cuda() for t in (x, ei, batch, ptr, sel, tgt)]
        _, g_plain = _step(hip, data)
        sync = GradSync(hip.parameters(), check=True, single_rank_collectives=True)
        sync.enable_overlap()
        _, g_stage = _step(hip, data)
        assert len(sync._segments) == 2 and all(w is not None for _, _, w in sync._segments)
        n = sync.all_reduce()
        torch.cuda.synchronize()
        assert n == sum(p.numel() for p in hip.parameters() if p.grad is not None) and not sync._segments
        for k, p in hip.named_parameters():
            if g_stage[k] is not None:
                assert torch.equal(p.grad, g_stage[k]), k               # AVG over one rank: unchanged
                assert (p.grad - g_plain[k]).abs().max().item() < 2e-6 * max(1.0, g_plain[k].abs().max().item())
        sync.enable_overlap(False)
        _, g_one = _step(hip, data)                                     # one bucket, SUM + 1/world
        assert sync.all_reduce() == n
        torch.cuda.synchronize()
        for k, p in hip.named_parameters():
            if g_one[k] is not None:
                assert torch.equal(p.grad, g_one[k]), k
        # HIP-graph capture of the step while the RCCL group is alive (bench.py captures before it creates the group; a
        # training script may not), then replay + the one-bucket all-reduce on the graph's gradient buffers
        from gnn_hex_amd.graphs import GraphedStep
        params = list(hip.parameters())
        xh = ops.attach_hints(data[0], is_maker=True, max_nodes=int((ptr[1:] - ptr[:-1]).max()))   # no host sync in a capture
        data[1]._hex_grouped = True

        def fn():
            for p in params:
                p.grad = None
            q = hip(xh, data[1], data[2], data[3])
            loss = torch.nn.functional.mse_loss(q[data[4]], data[5])
            loss.backward()
            return loss

        gstep = GraphedStep(fn, params)
        for _ in range(2):
            gstep.replay()
            assert sync.all_reduce() == n
        torch.cuda.synchronize()
        for k, p in hip.named_parameters():
            if g_one[k] is not None:
                assert torch.equal(p.grad, g_one[k]), k
        # round 4: the step as TWO captured graphs split at the staged backward's hand-over (what `bench.py --gpus N` replays):
        # replay 1, all-reduce of the finished tail segment (async, RCCL's stream), replay 2 beside it, all-reduce of the
        # head segment, wait -- gradients equal the one-graph step's, bit for bit (AVG over one rank)
        from gnn_hex_amd.graphs import GraphedSplitStep

        def first():
            for p in params:
                p.grad = None
            loss, _, _, call = ops.td_step(hip, xh, data[1], data[2], data[3], sel=data[4], target=data[5], defer_lower=True)
            return loss, call

        sstep = GraphedSplitStep(first, ops.finish_backward, params)
        assert sstep.flat is not None and 0 < sstep.cut < sstep.total == n
        for _ in range(2):
            sstep.replay_first()
            sync.reduce_segment(sstep.flat, sstep.cut, sstep.total)
            sstep.replay_second()
            sync.reduce_segment(sstep.flat, 0, sstep.cut)
            assert len(sync._segments) == 2 and all(w is not None for _, _, w in sync._segments)
            assert sync.all_reduce() == n
        torch.cuda.synchronize()
        for k, p in hip.named_parameters():
            if g_one[k] is not None:
                assert torch.equal(p.grad, g_one[k]), k
        # --noisy_dqn=True with the overlap switched on: the advantage linear's effective weight mu + sigma * eps
        # is no leaf, its gradient must not be all-reduced in place inside the backward -- the staging is skipped and the
        # finished mu / sigma gradients go through the bucket.  One rank: gradients unchanged, no "more than one backward"
        from argparse import Namespace
        from gnn_hex_amd.models import get_pre_defined
        torch.manual_seed(64)
        noisy = get_pre_defined("modern_two_headed", Namespace(num_layers=4, hidden_channels=35, norm=False, noisy_dqn=True,
                                                               noisy_sigma0=0.5, num_head_layers=2)).cuda()
        _, g_ref = _step(noisy, data)
        nsync = GradSync(noisy.parameters(), check=True, single_rank_collectives=True)
        nsync.enable_overlap()
        _, g_n = _step(noisy, data)
        assert not nsync._segments                       # nothing was handed over from inside the backward
        nn_ = nsync.all_reduce()
        torch.cuda.synchronize()
        assert nn_ == sum(p.numel() for p in noisy.parameters() if p.grad is not None)
        assert g_n["maker_head.linear.weight_sigma"] is not None or g_n["breaker_head.linear.weight_sigma"] is not None
        for k, p in noisy.named_parameters():
            if g_n[k] is not None:
                assert torch.equal(g_n[k], g_ref[k]), k
                assert torch.equal(p.grad, g_n[k]), k
    finally:
        ops.set_grad_stage_hook(None)
        from gnn_hex_amd import _lib as _hl
        _hl.lib().hexgnn_stack_reserve_cus(0)            # (enable_overlap reserved CUs for the RCCL channels)
        dist.destroy_process_group()
