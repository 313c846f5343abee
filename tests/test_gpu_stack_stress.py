"""Stress test of the one-launch SAGE stack kernels' cross-workgroup hand-over (MI355X guide: "test every
hand-off under UNEVEN load, consumer L1-warm, checking every word").

The ragged MIX256 batch (Hex-5..13, graphs cut by the 128-row block boundaries) runs through a 17-layer stack at hidden 110:
* reference A: per-layer launches (no cross-workgroup hand-over inside a launch) -- values to fp32 rounding;
* reference B: the one-launch kernels on an idle chip, no skew -- every later run must equal it BIT FOR BIT (a stale or torn row
  changes bits);
* then many launches with a new pseudo-random per-block delay pattern each (hexgnn_debug_stack_mode: blocks reach a layer up to
  ~100 us apart) while a streaming kernel on a second stream keeps 64 CUs' memory pipes busy: every word of every layer's
  activation slab and of every parameter gradient is compared.
Also: the timeout path poisons the output of the SAME call and is reported by hexgnn_stack_status()."""
import pytest
import torch

from helpers import batch_tensors

pytestmark = pytest.mark.gpu

HIDDEN, LAYERS = 110, 17


class _Conv(torch.nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.lin_l = torch.nn.Linear(cin, cout, bias=True)
        self.lin_r = torch.nn.Linear(cin, cout, bias=False)


def _case():
    torch.manual_seed(1)
    x, ei, batch, ptr = batch_tensors("D0", [5 + (g % 9) for g in range(256)], maker=True)
    convs = torch.nn.ModuleList([_Conv(3 if i == 0 else HIDDEN, HIDDEN) for i in range(LAYERS)])
    with torch.no_grad():
        for c in convs[1:]:                 # keep the signal alive through 17 mean-aggregating layers
            c.lin_r.weight.mul_(2.0)
    up = torch.randn(x.shape[0], HIDDEN)
    return x, ei, convs, up


def _step(ops, xd, gs, convs, upd):
    for p in convs.parameters():
        p.grad = None
    y = ops.sage_stack(xd, gs, 3, HIDDEN, convs)
    (y * upd).sum().backward()
    return y.detach().clone(), [p.grad.detach().clone() for p in convs.parameters()], None


def test_hand_over_under_skew_and_load_every_word():
    from gnn_hex_amd import _lib, ops
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    x, ei, convs, up = _case()
    convs = convs.to(dev)
    xd, upd = x.to(dev), up.to(dev)
    gs = ops.GraphStructure(ei.to(dev), x.shape[0])
    n = x.shape[0]
    assert (n + 127) // 128 >= 170, "the batch must span many row blocks"
    try:
        _lib.check(L.hexgnn_debug_stack_mode(0, 0))
        y_a, g_a, _ = _step(ops, xd, gs, convs, upd)
        _lib.check(L.hexgnn_debug_stack_mode(-1, 0))
        y_b, g_b, _ = _step(ops, xd, gs, convs, upd)
        torch.cuda.synchronize()
        assert L.hexgnn_stack_status(1) == 0
        scale = max(1.0, y_a.abs().max().item())
        assert y_a.abs().max().item() > 1e-3
        assert (y_b - y_a).abs().max().item() < 2e-5 * scale
        for a, b in zip(g_b, g_a):
            assert (a - b).abs().max().item() < 5e-5 * max(1.0, b.abs().max().item())
        # uneven load: skewed blocks + a streaming kernel on 64 CUs of a second stream
        big = torch.empty(256 * 1024 * 1024 // 4, dtype=torch.float32, device=dev).normal_()
        sink = torch.zeros(16, dtype=torch.float32, device=dev)
        side = torch.cuda.Stream()
        launches = 0
        for rep in range(130):
            _lib.check(L.hexgnn_debug_stack_mode(-1, 1000 + rep))
            if rep % 2 == 0:
                with torch.cuda.stream(side):
                    _lib.check(L.hexgnn_debug_occupy(64, 4000, big.data_ptr(), big.numel() * 4, sink.data_ptr(),
                                                     side.cuda_stream))
            y, g, _ = _step(ops, xd, gs, convs, upd)
            launches += 2                    # one forward + one backward stack launch per step
            assert torch.equal(y, y_b), "forward rows differ under skew (repeat %d)" % rep
            for k, (a, b) in enumerate(zip(g, g_b)):
                assert torch.equal(a, b), "gradient %d differs under skew (repeat %d)" % (k, rep)
        torch.cuda.synchronize()
        assert L.hexgnn_stack_status(1) == 0
        assert launches >= 260
    finally:
        L.hexgnn_debug_stack_mode(-1, 0)


def test_every_layer_slab_under_skew():
    """hexgnn_sage_stack_forward called directly: ALL 17 activation slabs and the saved aggregates, bit for bit, 250 skewed
    launches against the unskewed one (the autograd wrapper above only exposes the top layer and the gradients)."""
    import ctypes as C
    from gnn_hex_amd import _lib, ops
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    x, ei, convs, _ = _case()
    convs = convs.to(dev)
    xd = x.to(dev)[:, :3].contiguous()
    n = x.shape[0]
    gs = ops.GraphStructure(ei.to(dev), n)
    hp = ops.padded_width(HIDDEN)
    wl = (C.c_void_p * LAYERS)(*[c.lin_l.weight.data_ptr() for c in convs])
    bl = (C.c_void_p * LAYERS)(*[c.lin_l.bias.data_ptr() for c in convs])
    wr = (C.c_void_p * LAYERS)(*[c.lin_r.weight.data_ptr() for c in convs])
    wpack = torch.empty(L.hexgnn_sage_stack_pack_bytes(3, HIDDEN, LAYERS), dtype=torch.uint8, device=dev)
    saved_bytes = L.hexgnn_sage_stack_saved_bytes(n, 3, HIDDEN, LAYERS)

    def run():
        acts = torch.full((LAYERS, n, hp), float("nan"), dtype=torch.float32, device=dev)
        saved = torch.zeros(saved_bytes, dtype=torch.uint8, device=dev)
        _lib.check(L.hexgnn_sage_stack_forward(n, 3, HIDDEN, LAYERS, gs.rowptr.data_ptr(), gs.col.data_ptr(),
                                               gs.invdeg.data_ptr(), xd.data_ptr(), 3, wl, bl, wr, wpack.data_ptr(),
                                               acts.data_ptr(), saved.data_ptr(), 1, 0, ops._stream()))
        return acts, saved
    try:
        _lib.check(L.hexgnn_debug_stack_mode(-1, 0))
        a0, s0 = run()
        torch.cuda.synchronize()
        assert not torch.isnan(a0).any().item()
        big = torch.empty(256 * 1024 * 1024 // 4, dtype=torch.float32, device=dev).normal_()
        sink = torch.zeros(16, dtype=torch.float32, device=dev)
        side = torch.cuda.Stream()
        for rep in range(250):
            _lib.check(L.hexgnn_debug_stack_mode(-1, 5000 + rep))
            if rep % 3 == 0:
                with torch.cuda.stream(side):
                    _lib.check(L.hexgnn_debug_occupy(64, 3000, big.data_ptr(), big.numel() * 4, sink.data_ptr(),
                                                     side.cuda_stream))
            a, s = run()
            assert torch.equal(a, a0), "activation slabs differ under skew (repeat %d, first bad layer %d)" % (
                rep, int((a != a0).flatten(1).any(1).nonzero()[0]))
            assert torch.equal(s, s0), "saved aggregates differ under skew (repeat %d)" % rep
        torch.cuda.synchronize()
        assert L.hexgnn_stack_status(1) == 0
    finally:
        L.hexgnn_debug_stack_mode(-1, 0)


def test_second_stream_in_flight_falls_back_to_per_layer_launches():
    """While a one-launch kernel of this process is in flight on ANOTHER stream the guard keeps a new stack call on the per-layer
    launches (two resident-set kernels could starve each other); results stay the one-launch ones to fp32 rounding, and nothing
    times out."""
    from gnn_hex_amd import _lib, ops
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    x, ei, convs, up = _case()
    convs = convs.to(dev)
    xd, upd = x.to(dev), up.to(dev)
    gs = ops.GraphStructure(ei.to(dev), x.shape[0])
    y0, g0, _ = _step(ops, xd, gs, convs, upd)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for st in (s1, s2, s1, s2):
        with torch.cuda.stream(st):
            st.wait_stream(torch.cuda.current_stream())
            with torch.no_grad():
                outs.append(ops.sage_stack(xd, gs, 3, HIDDEN, convs))
    torch.cuda.synchronize()
    assert L.hexgnn_stack_status(1) == 0
    scale = max(1.0, y0.abs().max().item())
    for y in outs:
        assert (y - y0).abs().max().item() < 2e-5 * scale


def test_timeout_poisons_the_same_call_and_is_reported():
    """A wait that cannot end (test aid: one block never publishes its progress) must not compute on: the waves that needed its
    rows give up after their poll budget, write NaN rows from then on -- the output of the SAME call cannot pass for a result --
    and HEXGNN_ETIMEOUT is there for the status query (GraphStructure.check raises it).  Afterwards the library works again."""
    from gnn_hex_amd import _lib, ops
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    x, ei, convs, up = _case()
    convs = convs.to(dev)
    xd = x.to(dev)
    gs = ops.GraphStructure(ei.to(dev), x.shape[0])
    with torch.no_grad():
        y0 = ops.sage_stack(xd, gs, 3, HIDDEN, convs)
        torch.cuda.synchronize()
        assert L.hexgnn_stack_status(1) == 0
        try:
            _lib.check(L.hexgnn_debug_stack_mode(-1, 0xDE000000 | 40))       # block 40 never publishes
            y = ops.sage_stack(xd, gs, 3, HIDDEN, convs)
            torch.cuda.synchronize()
        finally:
            L.hexgnn_debug_stack_mode(-1, 0)
        assert torch.isnan(y).any().item(), "the call that timed out returned finite rows everywhere"
        # (the readers of the muted block give up, and -- all budgets being equal -- so do the waves that were waiting for THEIR
        # progress meanwhile: a neighbourhood of blocks around it, not the whole batch)
        bad_blocks = torch.unique(torch.isnan(y).any(1).nonzero().flatten() // 128)
        assert int(bad_blocks.min()) >= 20 and int(bad_blocks.max()) <= 60 and ((bad_blocks == 39) | (bad_blocks == 41)).any().item()
        with pytest.raises(_lib.HexGnnError):
            gs.check()
        assert L.hexgnn_stack_status(0) == 0        # reported once, cleared
        y1 = ops.sage_stack(xd, gs, 3, HIDDEN, convs)
        torch.cuda.synchronize()
        assert torch.equal(y1, y0)
