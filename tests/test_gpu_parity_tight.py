"""Parity gates that bite ("Tighten the parity gates").

The absolute 1e-4 bar of tests/test_gpu_fullsize.py is loose exactly where the freshly initialised network's signal is
small: with the default init the 15-layer mean-aggregating ReLU stack over-smooths (Q std 1.5e-5 over a batch,
advantage-linear gradients ~1e-5, first-layer gradients ~1e-3), so a 7 % error in those tensors -- or a zeroed
`A - mean(A)` -- would pass.  Here, for every benchmark batch (GNN-L Hex-11 / GNN-S Hex-7 / the ragged MIX batch, start
positions AND mid-game boards, 256 graphs) and every kernel path:

* ground truth = the SAME oracle evaluated in float64;
* every gradient tensor is held to a norm-RELATIVE bound: ||g_hip - g_64|| / ||g_64|| <= max(3 x the fp32 oracle's own
  distance from the float64 oracle, floor) -- two correct fp32 evaluations differ by accumulation order, three times that
  distance is still "an fp32 evaluation of the same arithmetic", anything structurally wrong is orders of magnitude out;
* two weight states:
  - the default init (collapsed: e.g. the advantage-linear gradient is a sum over nodes of (dq_i - mean dq) h_i with
    nearly identical h_i, a result ~1e-3 of its summands, so ANY fp32 evaluation carries a relative error of 1e-3..2e-2
    there -- measured on MI355X: fp32 oracle 1.2e-3 / 1.6e-3, HIP 5.1e-3 (L256-D0) / 1.9e-2 (S256-D1, layer-major) on the
    advantage linear): floor 5e-2, which a 7 % error or a zeroed tensor still fails;
  - a SHARPENED state (tests/helpers.py::sharpen_: high-pass SAGE layers, de-saturated value head) with
    std(A - mean A) >= 0.1 and every gradient tensor's |g|max >= 1e-2, asserted here on the oracle.  Floor 2e-3: a ReLU
    network is only piecewise smooth, and a pre-activation within rounding of zero flips its mask between two fp32
    evaluations with different summation orders -- one flipped element moves a gradient tensor by 1e-4..1e-3 of its norm
    (measured: S256-D1 6.0e-4 on BOTH exact-fp32 kernel paths, which share their fmaf chains, against 7e-6 for the fp32
    oracle and 2.3e-5 for the f16x3 path, whose roundings flip other elements; L256-D1 1.5e-4 vs 2.9e-5); tensors without
    a flip sit at 1e-6..7e-5, within 3x the fp32 oracle's own distance;
* Q itself: max |Q - Q_64| <= max(3 x the fp32 oracle's, 5e-6) (|Q| up to 2: 5e-6 is ~20 ulp after 17 layers; measured
  1e-7..2.4e-6), and on the sharpened state additionally the structural
  identities of GN0/models.py:571-584 on the device result (per-graph mean of Q == tanh(value)).

Also here: run-to-run bit reproducibility of the LAYER-MAJOR kernels (MIX, Hex-12+, --norm and hidden 113-128 all run
there) at the benchmark batches, 12 repeats -- the fused kernels' twin lives in tests/test_gpu_model.py.
"""
import copy

import pytest
import torch

from helpers import batch_tensors, make_pair, model_args, sel_and_targets, sharpen_

pytestmark = pytest.mark.gpu

MIX = [5 + (g % 9) for g in range(256)]
CASES = {
    # name: (num_layers, hidden, kind, sizes, maker)
    "L256-D0": (15, 110, "D0", [11] * 256, True),
    "L256-D1": (15, 110, "D1", [11] * 256, False),
    "S256-D0": (10, 35, "D0", [7] * 256, False),
    "S256-D1": (10, 35, "D1", [7] * 256, True),
    "MIX256-D0": (15, 110, "D0", MIX, False),
    "MIX256-D1": (15, 110, "D1", MIX, True),
}
_cache = {}


GSCALE = {"default": 1.0, "sharp": 64.0}     # upstream gradient scale: the 256-graph mean divides every gradient by 256;
#                                               with 64 x loss the sharpened state's smallest tensor has |g|max >= 1e-2


def _run(model, x, ei, batch, ptr, sel, tgt, gscale=1.0):
    model.zero_grad(set_to_none=True)
    q = model(x, ei, batch, ptr)
    (torch.nn.functional.mse_loss(q[sel], tgt) * gscale).backward()
    return q.detach(), {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in model.named_parameters()}


def _oracle(name, state):
    """fp32 and float64 oracle results of one (batch, weight state), computed once per session."""
    key = (name, state)
    if key not in _cache:
        layers, hidden, kind, sizes, maker = CASES[name]
        _, ref = make_pair(layers, hidden, seed=0, device="cpu")
        if state == "sharp":
            sharpen_(ref)
        x, ei, batch, ptr = batch_tensors(kind, sizes, maker=maker)
        sel, tgt = sel_and_targets(ptr)
        ref64 = copy.deepcopy(ref).double()     # (before the fp32 run: the oracle keeps final_conv_acts, a non-leaf tensor)
        q32, g32 = _run(ref, x, ei, batch, ptr, sel, tgt, GSCALE[state])
        q64, g64 = _run(ref64, x.double(), ei, batch, ptr, sel, tgt.double(), GSCALE[state])
        with torch.no_grad():
            v64, a64 = ref64(x.double(), ei, batch, ptr, seperate=True)
        _cache[key] = dict(state=ref.state_dict(), inputs=(x, ei, batch, ptr, sel, tgt), q32=q32, g32=g32, q64=q64,
                           g64=g64, v64=v64, a64=a64)
    return _cache[key]


@pytest.fixture(params=[(True, "fp32"), (True, "f16x3"), (False, "fp32")], ids=["fused", "fused-f16x3", "layered"])
def path(request):
    from gnn_hex_amd import ops
    ops.set_fused(request.param[0])
    ops.set_math(request.param[1])
    yield request.param
    ops.set_fused(True)
    ops.set_math("fp32")


@pytest.mark.parametrize("state", ["default", "sharp"])
@pytest.mark.parametrize("name", sorted(CASES))
def test_relative_parity_against_float64_oracle(name, state, path):
    from gnn_hex_amd.models import get_pre_defined
    layers, hidden, kind, sizes, maker = CASES[name]
    if name.startswith("MIX") and path[0]:
        pytest.skip("graphs above 128 nodes: the batch runs on the layer-major kernels whatever the switch says")
    o = _oracle(name, state)
    x, ei, batch, ptr, sel, tgt = o["inputs"]
    if state == "sharp":
        # the point of this state: the oracle's own signal must not collapse
        assert o["a64"].std().item() >= 0.1, "std(A - mean A) = %g" % o["a64"].std().item()
        assert (o["q64"].max() - o["q64"].min()).item() >= 0.5
        for k, g in o["g64"].items():
            if g is not None:
                assert g.abs().max().item() >= 1e-2, "%s: |g|max %g" % (k, g.abs().max().item())
    hip = get_pre_defined("modern_two_headed", model_args(layers, hidden))
    hip.load_state_dict(o["state"])
    hip = hip.cuda()
    xd, eid = x.cuda(), ei.cuda()
    xd._hex_is_maker = maker
    xd._hex_max_nodes = int((ptr[1:] - ptr[:-1]).max())
    eid._hex_grouped = True
    q, g = _run(hip, xd, eid, batch.cuda(), ptr.cuda(), sel.cuda(), tgt.cuda(), GSCALE[state])
    torch.cuda.synchronize()
    split = path[1] == "f16x3"
    floor = 5e-2 if state == "default" else 2e-3

    eq = (q.cpu().double() - o["q64"]).abs().max().item()
    eq32 = (o["q32"].double() - o["q64"]).abs().max().item()
    assert eq <= max(3.0 * eq32, 8e-6 if split else 5e-6), "%s/%s: |Q - Q64| %g (fp32 oracle %g)" % (name, state, eq, eq32)

    worst = (0.0, 0.0, "")
    for k, g64 in o["g64"].items():
        if g64 is None:
            assert g[k] is None, k
            continue
        assert g[k] is not None, k
        nrm = g64.norm().item()
        if nrm < 1e-6:      # a tensor whose true gradient vanishes is rounding noise in any arithmetic: absolute bound
            assert (g[k].cpu().double() - g64).abs().max().item() < 1e-6, k
            continue
        rel = (g[k].cpu().double() - g64).norm().item() / nrm
        rel32 = (o["g32"][k].double() - g64).norm().item() / nrm
        if rel > worst[0]:
            worst = (rel, rel32, k)
        assert rel <= max(3.0 * rel32, floor), \
            "%s/%s %s: ||g - g64|| / ||g64|| = %.3g, the fp32 oracle's own %.3g, floor %g" % (name, state, k, rel, rel32, floor)
    print("%s %s %s: |Q-Q64| %.3g (oracle32 %.3g); worst gradient tensor %s rel %.3g (oracle32 %.3g)"
          % (name, state, path, eq, eq32, worst[2], worst[0], worst[1]))

    if state == "sharp":
        # structural identities on the DEVICE result (GN0/models.py:571-584): per graph mean_i Q_i == tanh(value)
        with torch.no_grad():
            v, a = hip(xd, eid, batch.cuda(), ptr.cuda(), seperate=True)
        torch.cuda.synchronize()
        p = ptr.tolist()
        qc = q.cpu().double()
        means = torch.stack([qc[p[i]:p[i + 1]].mean() for i in range(len(p) - 1)])
        assert (means - v.cpu().double()).abs().max().item() < 5e-6
        assert (v.cpu().double() - o["v64"]).abs().max().item() < (2e-5 if split else 5e-6)
        ea = (a.cpu().double() - o["a64"]).norm().item() / o["a64"].norm().item()
        assert ea < (1e-4 if split else 2e-5), "A - mean(A): relative error %g" % ea


@pytest.mark.parametrize("name", ["MIX256-D0", "L256-D1"])
def test_layer_major_path_is_bit_reproducible(name):
    """12 repeats of the benchmark batch on the LAYER-MAJOR kernels (sage_hidden_fwd/bwd_kernel, sage_dw_kernel, head and
    first-layer kernels): Q and every gradient bit-identical from run to run.  The kernels have no float atomics; what this
    guards is a timing-dependent hazard between MFMAs and the loads issued around them, which would
    show up as run-to-run differences under full-chip contention first."""
    from gnn_hex_amd import ops
    from gnn_hex_amd.models import get_pre_defined
    layers, hidden, kind, sizes, maker = CASES[name]
    o = _oracle(name, "sharp")
    x, ei, batch, ptr, sel, tgt = o["inputs"]
    hip = get_pre_defined("modern_two_headed", model_args(layers, hidden))
    hip.load_state_dict(o["state"])
    hip = hip.cuda()
    xd, eid = x.cuda(), ei.cuda()
    xd._hex_is_maker = maker
    xd._hex_max_nodes = int((ptr[1:] - ptr[:-1]).max())
    eid._hex_grouped = True
    args = (xd, eid, batch.cuda(), ptr.cuda(), sel.cuda(), tgt.cuda())
    ops.set_fused(False)
    try:
        q0, g0 = _run(hip, *args)
        for _ in range(12):
            q, g = _run(hip, *args)
            assert torch.equal(q, q0)
            for k in g0:
                assert (g[k] is None) == (g0[k] is None)
                if g0[k] is not None:
                    assert torch.equal(g[k], g0[k]), k
    finally:
        ops.set_fused(True)
