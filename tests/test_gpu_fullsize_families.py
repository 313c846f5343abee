"""The model families added in round 3 at the batch size every BASELINE configuration trains on (256 Hex-11 boards): the
two_headed family with CachedGraphNorm, the HexAra network, and a width beyond 128 -- against the fp32 oracle, with the
float64 oracle as the yardstick where ReLU masks of near-zero pre-activations differ between any two fp32 evaluations
(tests/test_gpu_norm.py::test_norm_model_at_the_benchmark_batch explains the criterion)."""
import copy
from argparse import Namespace

import pytest
import torch

from helpers import batch_tensors, make_pair, sel_and_targets, sharpen_

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _step(model, x, ei, batch, ptr, sel, tgt):
    model.zero_grad(set_to_none=True)
    q = model(x, ei, batch, ptr)
    torch.nn.functional.mse_loss(q.reshape(-1)[sel], tgt).backward()
    return q.detach(), {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in model.named_parameters()}


def _check_against_both(hip, ref, x, ei, batch, ptr, what):
    ref64 = copy.deepcopy(ref).double()
    sel, tgt = sel_and_targets(ptr)
    q_ref, g_ref = _step(ref, x, ei, batch, ptr, sel, tgt)
    q64, g64 = _step(ref64, x.double(), ei, batch, ptr, sel, tgt.double())
    q, g = _step(hip, *[t.cuda() for t in (x, ei, batch, ptr, sel, tgt)])
    torch.cuda.synchronize()
    eq = (q.cpu().double() - q64).abs().max().item()
    er = (q_ref.double() - q64).abs().max().item()
    assert (q.cpu() - q_ref).abs().max().item() < TOL or eq < 3.0 * er, "%s: Q vs float64 %g (fp32 oracle %g)" % (what, eq, er)
    worst = 0.0
    for k in g_ref:
        if g_ref[k] is None:
            assert g[k] is None or float(g[k].abs().max()) == 0.0, k
            continue
        scale = max(1.0, g_ref[k].abs().max().item())
        e32 = (g[k].cpu() - g_ref[k]).abs().max().item() / scale
        eh = (g[k].cpu().double() - g64[k]).abs().max().item() / scale
        er = (g_ref[k].double() - g64[k]).abs().max().item() / scale
        worst = max(worst, eh)
        assert e32 < TOL or eh < 3.0 * er, "%s %s: vs fp32 oracle %g, vs float64 %g (fp32 oracle vs float64 %g)" % (what, k, e32, eh, er)
    print("%s: worst gradient distance from the float64 oracle %.3g" % (what, worst))


@pytest.mark.parametrize("norm", [False, True])
def test_two_headed_at_the_benchmark_batch(norm):
    from gnn_hex_amd.models import get_pre_defined
    from oracle.model_ref import get_pre_defined_ref
    args = Namespace(num_layers=13, hidden_channels=32, norm=norm, noisy_dqn=False, noisy_sigma0=0.5, num_head_layers=2)
    torch.manual_seed(17)
    ref = get_pre_defined_ref("two_headed", args)
    with torch.no_grad():
        for k, p in ref.named_parameters():
            if "norm" in k:
                p.add_(torch.randn(p.shape) * 0.2)
    hip = get_pre_defined("two_headed", args)
    hip.load_state_dict(ref.state_dict())
    x, ei, batch, ptr = batch_tensors("D1", [11] * 256, maker=True)
    _check_against_both(hip.cuda(), ref, x, ei, batch, ptr, "two_headed norm=%s" % norm)


def test_wide_model_at_the_benchmark_batch():
    hip, ref = make_pair(15, 160, seed=8)
    x, ei, batch, ptr = batch_tensors("D1", [11] * 256, maker=False)
    _check_against_both(hip, ref, x, ei, batch, ptr, "hidden 160")


@pytest.mark.parametrize("swap_allowed", [False, True])
def test_hexara_network_at_512_positions(swap_allowed):
    from gnn_hex_amd.torch_script_models import get_current_model
    from oracle.hexara_ref import get_current_model_ref
    torch.manual_seed(4)
    ref = get_current_model_ref(swap_allowed=swap_allowed)
    # a weight state whose signal survives 15 mean-aggregating layers (helpers.sharpen_): with the default init the policy
    # gradients are ~1e-7 per entry and sums of cancelling terms (the fp32 oracle itself sits 2 % from float64 there)
    sharpen_(ref)
    with torch.no_grad():
        for name in ("value_linear", "swap_linear"):
            ref.my_modules[name].layers[0].weight.mul_(0.02)
    hip = get_current_model(swap_allowed=swap_allowed)
    hip.load_state_dict(ref.state_dict())
    hip = hip.cuda()
    x, ei, batch, ptr = batch_tensors("D1", [11] * 512, maker=True)
    x = x.clone()
    for g in range(512):
        x[ptr[g]:ptr[g + 1], 2] = float(g % 3 == 0)
    ref64 = copy.deepcopy(ref).double()
    pi_r, v_r, gi_r, bp_r = ref(x, ei, batch, ptr)
    gen = torch.Generator().manual_seed(1)
    tp, tv = torch.rand(pi_r.numel(), generator=gen), torch.rand(512, generator=gen) * 2 - 1

    def loss_of(pi, v, tp, tv):
        return -(pi * tp).sum() / 512 + torch.nn.functional.mse_loss(v, tv)

    ref.zero_grad()
    loss_of(pi_r, v_r, tp, tv).backward()
    pi64, v64, _, _ = ref64(x.double(), ei, batch, ptr)
    loss_of(pi64, v64, tp.double(), tv.double()).backward()
    pi, v, gi, bp = hip(x.cuda(), ei.cuda(), batch.cuda(), ptr.cuda())
    hip.zero_grad()
    loss_of(pi, v, tp.cuda(), tv.cuda()).backward()
    torch.cuda.synchronize()
    assert torch.equal(gi.cpu(), gi_r) and torch.equal(bp.cpu(), bp_r)
    assert (pi.detach().cpu() - pi_r.detach()).abs().max().item() < 3e-4
    assert (v.detach().cpu() - v_r.detach()).abs().max().item() < TOL
    # gradients NORM-RELATIVE against the float64 oracle; the fp32 oracle's own distance from float64 is the yardstick (3x),
    # with a 0.5 % floor (ReLU masks of near-zero pre-activations differ between any two fp32 evaluations)
    g32, g64 = dict(ref.named_parameters()), dict(ref64.named_parameters())
    for k, p in hip.named_parameters():
        if g64[k].grad is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        nrm = g64[k].grad.norm().item()
        if nrm < 1e-8:      # e.g. the bias of the policy head's last layer: it shifts every logit of a graph alike, softmax is
            assert float(p.grad.abs().max()) < 1e-5, k          # invariant, the exact gradient is 0 and fp32 returns rounding noise
            continue
        rel_h = (p.grad.cpu().double() - g64[k].grad).norm().item() / nrm
        rel_r = (g32[k].grad.double() - g64[k].grad).norm().item() / nrm
        assert rel_h <= max(3.0 * rel_r, 5e-3), "%s: relative distance from float64 %g (fp32 oracle %g)" % (k, rel_h, rel_r)
