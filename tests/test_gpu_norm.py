"""--norm=True on the device: torch_geometric LayerNorm in the whole-batch "graph" form the model
uses (GN0/models.py:8,286-287,550-551,935,945).  The norm kernels against the torch expression (forward, input / weight /
bias gradients, with and without the fused ReLU), then the full modern_two_headed network with norms everywhere (body
layers, after_embed_norm, head layers) against the oracle at 1e-4, with non-trivial norm weights."""
from argparse import Namespace

import pytest
import torch

from helpers import batch_tensors, sel_and_targets

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _ref_norm(x, w, b, eps, relu):
    xc = x - x.mean()
    out = xc / (xc.std(unbiased=False) + eps) * w + b
    return torch.relu(out) if relu else out


@pytest.mark.parametrize("n,hidden", [(1, 16), (37, 35), (1000, 110), (5000, 128), (700, 24)])
@pytest.mark.parametrize("relu", [False, True])
def test_graph_layernorm_kernels_match_torch(n, hidden, relu):
    from gnn_hex_amd import ops
    gen = torch.Generator().manual_seed(n + hidden)
    x = torch.randn(n, hidden, generator=gen) * 3.0 + 1.5
    w = torch.rand(hidden, generator=gen) + 0.5
    b = torch.randn(hidden, generator=gen) * 0.3
    up = torch.randn(n, hidden, generator=gen)
    ref_in = [t.clone().double().requires_grad_(True) for t in (x, w, b)]
    y_ref = _ref_norm(*ref_in, 1e-5, relu)
    (y_ref * up.double()).sum().backward()
    dev_in = [t.clone().cuda().requires_grad_(True) for t in (x, w, b)]
    y = ops.graph_layernorm(*dev_in, 1e-5, relu)
    (y * up.cuda()).sum().backward()
    torch.cuda.synchronize()
    assert y.shape == (n, hidden)
    assert (y.cpu().double() - y_ref).abs().max().item() < 2e-5
    for got, want, name in zip(dev_in, ref_in, ("x", "weight", "bias")):
        scale = max(1.0, want.grad.abs().max().item())
        assert (got.grad.cpu().double() - want.grad).abs().max().item() < 2e-5 * scale, name
    # deterministic: fixed-shape reductions
    y2 = ops.graph_layernorm(dev_in[0].detach(), dev_in[1].detach(), dev_in[2].detach(), 1e-5, relu)
    assert torch.equal(y2, y.detach())


def _norm_pair(layers, hidden, seed, noisy=False):
    from gnn_hex_amd.models import get_pre_defined
    from oracle.model_ref import get_pre_defined_ref
    args = Namespace(num_layers=layers, hidden_channels=hidden, norm=True, noisy_dqn=noisy, noisy_sigma0=0.5,
                     num_head_layers=2)
    torch.manual_seed(seed)
    ref = get_pre_defined_ref("modern_two_headed", args)
    with torch.no_grad():                          # non-trivial affine parameters (the default is weight 1, bias 0)
        for k, p in ref.named_parameters():
            if "norm" in k:
                p.add_(torch.randn(p.shape) * 0.2)
    hip = get_pre_defined("modern_two_headed", args)
    missing = hip.load_state_dict(ref.state_dict())
    assert not missing.missing_keys and not missing.unexpected_keys
    return hip.cuda(), ref


def _step(model, x, ei, batch, ptr, sel, tgt, **kw):
    model.zero_grad(set_to_none=True)
    q = model(x, ei, batch, ptr, **kw)
    torch.nn.functional.mse_loss(q.reshape(-1)[sel], tgt).backward()
    return q.detach(), {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in model.named_parameters()}


@pytest.mark.parametrize("layers,hidden,sizes,noisy", [(3, 35, [7, 5, 9, 7], False), (6, 110, [11, 7, 11, 9, 5], False),
                                                       (3, 48, [5, 7, 13, 6], True)])
@pytest.mark.parametrize("maker", [True, False])
def test_modern_two_headed_with_norm_matches_oracle(layers, hidden, sizes, noisy, maker):
    hip, ref = _norm_pair(layers, hidden, seed=71, noisy=noisy)
    assert hip.after_embed_norm is not None and len(hip.gnn.norms) == layers and len(hip.maker_head.gnn.norms) == 2
    x, ei, batch, ptr = batch_tensors("D1", sizes, maker=maker)
    sel, tgt = sel_and_targets(ptr)
    dev = [t.cuda() for t in (x, ei, batch, ptr, sel, tgt)]
    for kw in ({}, {"advantages_only": True}):
        q_ref, g_ref = _step(ref, x, ei, batch, ptr, sel, tgt, **kw)
        q, g = _step(hip, *dev, **kw)
        torch.cuda.synchronize()
        assert q.shape == q_ref.shape and (q.cpu() - q_ref).abs().max().item() < TOL
        for k in g_ref:
            if g_ref[k] is None:
                assert g[k] is None, k
            else:
                err = (g[k].cpu() - g_ref[k]).abs().max().item()
                assert err < TOL * max(1.0, g_ref[k].abs().max().item()), "%s grad err %g" % (k, err)
    assert (hip.final_conv_acts.cpu() - ref.final_conv_acts).abs().max().item() < TOL       # after after_embed_norm
    # the statistics are batch-global: the same graph in a different batch gives different Q (the reference's behaviour)
    with torch.no_grad():
        x1, ei1, b1, p1 = batch_tensors("D1", sizes[:1], maker=maker)
        q_single = hip(x1.cuda(), ei1.cuda(), b1.cuda(), p1.cuda())
        q_single_ref = ref(x1, ei1, b1, p1)
    assert (q_single.cpu() - q_single_ref).abs().max().item() < TOL
