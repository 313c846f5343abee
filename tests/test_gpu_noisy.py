"""--noisy_dqn=True on the device: the heads' advantage linear is a FactorizedNoisyLinear
(GN0/models.py:84-141,331-334).  With identical parameters AND identical noise buffers the HIP path must match the oracle
at 1e-4 on Q and on every gradient (weight_mu / weight_sigma / bias_mu / bias_sigma included), on all kernel paths;
reset_noise changes the output, disable_noise reduces it to the mu network."""
from argparse import Namespace

import pytest
import torch

from helpers import batch_tensors, sel_and_targets

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


def _noisy_pair(layers, hidden, seed):
    from gnn_hex_amd.models import get_pre_defined
    from oracle.model_ref import get_pre_defined_ref
    args = Namespace(num_layers=layers, hidden_channels=hidden, norm=False, noisy_dqn=True, noisy_sigma0=0.5,
                     num_head_layers=2)
    torch.manual_seed(seed)
    ref = get_pre_defined_ref("modern_two_headed", args)
    hip = get_pre_defined("modern_two_headed", args)
    hip.load_state_dict(ref.state_dict())            # parameters and the epsilon buffers
    return hip.cuda(), ref


def _step(model, x, ei, batch, ptr, sel, tgt, **kw):
    model.zero_grad(set_to_none=True)
    q = model(x, ei, batch, ptr, **kw)
    torch.nn.functional.mse_loss(q.reshape(-1)[sel], tgt).backward()
    return q.detach(), {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in model.named_parameters()}


@pytest.mark.parametrize("maker", [True, False])
def test_noisy_heads_match_oracle(maker):
    hip, ref = _noisy_pair(4, 35, seed=51)
    assert {"maker_head.linear.weight_mu", "maker_head.linear.weight_sigma", "maker_head.linear.bias_epsilon"} \
        <= set(hip.state_dict())
    x, ei, batch, ptr = batch_tensors("D1", [7, 5, 9, 7, 11], maker=maker)
    sel, tgt = sel_and_targets(ptr)
    dev = [t.cuda() for t in (x, ei, batch, ptr, sel, tgt)]
    for kw in ({}, {"advantages_only": True}):
        q_ref, g_ref = _step(ref, x, ei, batch, ptr, sel, tgt, **kw)
        q, g = _step(hip, *dev, **kw)
        torch.cuda.synchronize()
        assert q.shape == q_ref.shape and (q.cpu() - q_ref).abs().max().item() < TOL
        head = "maker_head" if maker else "breaker_head"
        assert g_ref[head + ".linear.weight_sigma"] is not None and g_ref[head + ".linear.weight_sigma"].abs().max() > 0
        for k in g_ref:
            if g_ref[k] is None:
                assert g[k] is None, k
            else:
                err = (g[k].cpu() - g_ref[k]).abs().max().item()
                assert err < TOL * max(1.0, g_ref[k].abs().max().item()), "%s grad err %g" % (k, err)


def test_reset_and_disable_noise():
    hip, ref = _noisy_pair(3, 24, seed=52)
    x, ei, batch, ptr = batch_tensors("D0", [5, 7])
    dev = [t.cuda() for t in (x, ei, batch, ptr)]
    with torch.no_grad():
        q0 = hip(*dev)
        hip.maker_head.linear.reset_noise()
        q1 = hip(*dev)
        assert (q1 - q0).abs().max().item() > 1e-6                     # fresh noise, different advantages
        hip.maker_head.linear.disable_noise()
        ref.maker_head.linear.weight_epsilon.zero_()
        ref.maker_head.linear.bias_epsilon.zero_()
        q2, q2_ref = hip(*dev), ref(x, ei, batch, ptr)
    assert (q2.cpu() - q2_ref).abs().max().item() < TOL
