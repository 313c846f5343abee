"""Curriculum growth and checkpoints ON THE DEVICE: after ``grow_depth`` / ``grow_width``
(GN0/models.py:166-238,336-360,494-508) the HIP path must (a) still compute the function it computed before -- identity
layers preserve Q, zero-padded widening preserves the advantage stream (the value MLP is widened by a plain block copy,
GN0/models.py:51-73, which moves the [sum|max|min|mean] segments of its input: the reference's value output changes
there, and the mirror reproduces exactly that) -- and (b) agree with an oracle network BUILT at the grown shape and loaded with the grown state dict: Q and every gradient at
1e-4, on the fused (both arithmetic modes) and the layer-major kernels.  A checkpoint in the reference's format
({"state_dict", "args"}) written from the grown model loads into a fresh mirror and runs."""
import pytest
import torch

from helpers import batch_tensors, make_pair, model_args, sel_and_targets

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(params=[(True, "fp32"), (True, "f16x3"), (False, "fp32")], ids=["fused", "fused-f16x3", "layered"],
                autouse=True)
def _all_paths(request):
    from gnn_hex_amd import ops
    ops.set_fused(request.param[0])
    ops.set_math(request.param[1])
    yield
    ops.set_fused(True)
    ops.set_math("fp32")


def _step(model, x, ei, batch, ptr, sel, tgt):
    model.zero_grad(set_to_none=True)
    q = model(x, ei, batch, ptr)
    torch.nn.functional.mse_loss(q[sel], tgt).backward()
    return q.detach(), {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in model.named_parameters()}


def _oracle_like(hip, layers, hidden):
    from oracle.model_ref import get_pre_defined_ref
    ref = get_pre_defined_ref("modern_two_headed", model_args(layers, hidden))
    missing = ref.load_state_dict({k: v.cpu() for k, v in hip.state_dict().items()})
    assert not missing.missing_keys and not missing.unexpected_keys
    return ref


def _parity(hip, ref, data):
    x, ei, batch, ptr, sel, tgt = data
    q_ref, g_ref = _step(ref, x, ei, batch, ptr, sel, tgt)
    q, g = _step(hip, *[t.cuda() for t in data])
    torch.cuda.synchronize()
    assert (q.cpu() - q_ref).abs().max().item() < TOL
    for k in g_ref:
        if g_ref[k] is None:
            assert g[k] is None, k
        else:
            err = (g[k].cpu() - g_ref[k]).abs().max().item()
            assert err < TOL * max(1.0, g_ref[k].abs().max().item()), "%s grad err %g" % (k, err)
    return q


def _adv(hip, data):
    with torch.no_grad():
        return hip(*[t.cuda() for t in data[:4]], advantages_only=True)


def _data(sizes, maker):
    x, ei, batch, ptr = batch_tensors("D1", sizes, maker=maker)
    sel, tgt = sel_and_targets(ptr)
    return x, ei, batch, ptr, sel, tgt


@pytest.mark.parametrize("maker", [True, False])
def test_grow_depth_then_width_on_device(maker):
    hip, ref = make_pair(4, 35, seed=41)
    data = _data([7, 5, 11, 7, 9, 6], maker)
    q0 = _parity(hip, ref, data)
    a0 = _adv(hip, data)
    # depth: + 2 identity layers (lin_l = 0, lin_r = I) -> same function, deeper network
    hip.grow_depth(2)
    assert len(hip.gnn.convs) == 6
    q1 = _parity(hip, _oracle_like(hip, 6, 35), data)
    assert (q1 - q0).abs().max().item() < 1e-5
    # width 35 -> 48: old weights in the top-left blocks, new input columns zero -> same function, wider network
    hip.grow_width(48)
    assert hip.gnn.hidden_channels == 48 and hip.maker_head.linear.weight.shape == (1, 48)
    _parity(hip, _oracle_like(hip, 6, 48), data)
    assert (_adv(hip, data) - a0).abs().max().item() < 1e-5
    # width 48 -> 128: beyond the fused kernels' 112 columns, the layer-major kernels take over on every path
    hip.grow_width(128)
    q3 = _parity(hip, _oracle_like(hip, 6, 128), data)
    assert (_adv(hip, data) - a0).abs().max().item() < 1e-5
    # training continues on the grown network: one SGD step moves the output, parity still holds afterwards
    ref3 = _oracle_like(hip, 6, 128)
    for m in (hip, ref3):
        _step(m, *([t.cuda() for t in data] if m is hip else data))
        with torch.no_grad():
            for p in m.parameters():
                if p.grad is not None:
                    p.add_(p.grad, alpha=-0.05)
    q4 = _parity(hip, ref3, data)
    assert (q4 - q3).abs().max().item() > 1e-6


def test_gnn_l_grow_width_110_to_128():
    """The BASELINE GNN-L shape growing past the fused kernels' width (hidden 110 -> 128)."""
    hip, ref = make_pair(15, 110, seed=42)
    data = _data([11, 11, 7, 11], True)
    _parity(hip, ref, data)
    a0 = _adv(hip, data)
    hip.grow_width(128)
    _parity(hip, _oracle_like(hip, 15, 128), data)
    assert (_adv(hip, data) - a0).abs().max().item() < 1e-5


def test_checkpoint_of_a_grown_model_loads_and_runs(tmp_path):
    from gnn_hex_amd.models import get_pre_defined
    hip, _ = make_pair(3, 24, seed=43)
    hip.grow_depth(1)
    hip.grow_width(40)
    args = model_args(4, 40)
    path = str(tmp_path / "checkpoint_grown.pt")
    torch.save({"state_dict": hip.state_dict(), "args": args, "cache": None}, path)     # evaluate_elo.py:98-102 format
    stuff = torch.load(path, weights_only=False)
    fresh = get_pre_defined("modern_two_headed", args=stuff["args"])
    missing = fresh.load_state_dict(stuff["state_dict"])
    assert not missing.missing_keys and not missing.unexpected_keys
    fresh = fresh.cuda()
    if stuff["cache"] is not None:
        fresh.import_norm_cache(*stuff["cache"])
    data = _data([7, 5, 9], False)
    q_a, g_a = _step(hip, *[t.cuda() for t in data])
    q_b, g_b = _step(fresh, *[t.cuda() for t in data])
    assert torch.equal(q_a, q_b)
    for k in g_a:
        assert (g_a[k] is None) == (g_b[k] is None) and (g_a[k] is None or torch.equal(g_a[k], g_b[k])), k
    _parity(fresh, _oracle_like(fresh, 4, 40), data)


def test_grow_depth_and_width_with_norms():
    """--grow together with --norm=True (GN0/models.py:180-183, 226-235, 499-508): grow_depth appends a fresh norm per new
    layer, grow_width widens every norm (body, both heads, after_embed_norm) with the old affine parameters in front.  The
    whole-batch LayerNorm's statistics change with the width, so the FUNCTION is not preserved (in the reference neither);
    what must hold is parity with an oracle network built at the grown shape and loaded with the grown state dict."""
    from argparse import Namespace
    from gnn_hex_amd import ops
    from gnn_hex_amd.models import get_pre_defined
    from oracle.model_ref import get_pre_defined_ref
    if ops._FUSED_ENABLED and ops.get_math() != "fp32":
        pytest.skip("norm models run on the layer-major kernels whatever the switches say: once is enough")

    def nargs(layers, hidden):
        return Namespace(num_layers=layers, hidden_channels=hidden, norm=True, noisy_dqn=False, noisy_sigma0=0.5,
                         num_head_layers=2)

    torch.manual_seed(44)
    hip = get_pre_defined("modern_two_headed", nargs(3, 24))
    with torch.no_grad():
        for k, p in hip.named_parameters():
            if "norm" in k:
                p.add_(torch.randn(p.shape) * 0.2)
    hip = hip.cuda()

    def oracle(layers, hidden):
        ref = get_pre_defined_ref("modern_two_headed", nargs(layers, hidden))
        missing = ref.load_state_dict({k: v.cpu() for k, v in hip.state_dict().items()})
        assert not missing.missing_keys and not missing.unexpected_keys
        return ref

    data = _data([7, 5, 9, 6], True)
    _parity(hip, oracle(3, 24), data)
    hip.grow_depth(2)
    assert len(hip.gnn.convs) == 5 and len(hip.gnn.norms) == 5
    _parity(hip, oracle(5, 24), data)
    old_w = hip.gnn.norms[1].weight.detach().clone()
    hip.grow_width(40)
    assert hip.gnn.norms[1].weight.shape == (40,) and torch.equal(hip.gnn.norms[1].weight[:24], old_w)
    assert hip.after_embed_norm.weight.shape == (40,) and hip.maker_head.gnn.norms[0].weight.shape == (40,)
    assert torch.all(hip.gnn.norms[1].weight[24:] == 1) and torch.all(hip.gnn.norms[1].bias[24:] == 0)
    _parity(hip, oracle(5, 40), data)
