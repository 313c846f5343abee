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


def test_norm_model_at_the_benchmark_batch():
    """GNN-L with --norm=True on 256 Hex-11 mid-game boards (the batch size every BASELINE configuration trains on): the
    whole-stack call (one batched weight-gradient GEMM over 15 + 2 layers, 512-block statistics) against the oracle.
    LayerNorm centres the activations, so some of the 3.5 M pre-ReLU values per layer sit within rounding of zero and their
    masks differ between ANY two fp32 evaluations; the fp32 oracle itself is 4e-5 off its float64 twin here.  Q is held to
    1e-4 against the fp32 oracle; every gradient to 1e-4 against the fp32 oracle OR to within 3x the fp32 oracle's own
    distance from the float64 oracle."""
    import copy
    hip, ref = _norm_pair(15, 110, seed=3, noisy=False)
    ref64 = copy.deepcopy(ref).double()
    x, ei, batch, ptr = batch_tensors("D1", [11] * 256, maker=True)
    sel, tgt = sel_and_targets(ptr)
    dev = [t.cuda() for t in (x, ei, batch, ptr, sel, tgt)]
    q_ref, g_ref = _step(ref, x, ei, batch, ptr, sel, tgt)
    _, g64 = _step(ref64, x.double(), ei, batch, ptr, sel, tgt.double())
    q, g = _step(hip, *dev)
    torch.cuda.synchronize()
    assert (q.cpu() - q_ref).abs().max().item() < TOL
    worst_hip = worst_ref = 0.0
    for k in g_ref:
        if g_ref[k] is None:
            assert g[k] is None, k
            continue
        scale = max(1.0, g_ref[k].abs().max().item())
        e32 = (g[k].cpu() - g_ref[k]).abs().max().item() / scale
        eh = (g[k].cpu().double() - g64[k]).abs().max().item() / scale
        er = (g_ref[k].double() - g64[k]).abs().max().item() / scale
        worst_hip, worst_ref = max(worst_hip, eh), max(worst_ref, er)
        assert e32 < TOL or eh < 3.0 * er, "%s: vs fp32 oracle %g, vs float64 %g (fp32 oracle vs float64 %g)" % (k, e32, eh, er)
    print("norm L256-D1 gradients vs the float64 oracle: HIP %.3g, fp32 oracle %.3g" % (worst_hip, worst_ref))


def test_device_rollout_refuses_norm_models():
    """the rollout hands the model capacity-sized buffers whose tail rows are stale after node removals; the
    whole-batch LayerNorm statistics would include them.  It must refuse instead of returning silently different Q-values."""
    from gnn_hex_amd.multi_env_manager import DeviceRollout, Env_manager
    hip, _ = _norm_pair(3, 35, seed=9, noisy=False)
    mgr = Env_manager(4, 5, device="cuda")
    mgr.reset()
    with pytest.raises(NotImplementedError, match="norm"):
        DeviceRollout(mgr, hip, steps=2, eps=0.0, graph=False)
