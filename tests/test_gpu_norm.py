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


def test_live_row_count_equals_exact_size_call():
    """hexgnn_graph_layernorm_forward_live / ops.live_rows: over a capacity-sized buffer with a device-side row count the norm
    is, bit for bit, the exact-size call on the live rows, and rows behind the count are not written."""
    from gnn_hex_amd import ops
    gen = torch.Generator().manual_seed(5)
    cap, hidden = 3000, 35
    x = (torch.randn(cap, hidden, generator=gen) * 2.0 + 0.7).cuda()
    x[2000:] = float("nan")                                     # stale rows: must not reach the statistics
    w, b = (torch.rand(hidden, generator=gen) + 0.5).cuda(), torch.randn(hidden, generator=gen).cuda()
    for live in (2000, 1, 777, 0):
        cnt = torch.tensor([live], dtype=torch.int32, device="cuda")
        with torch.no_grad(), ops.live_rows(cnt):
            got = ops.graph_layernorm(x, w, b, 1e-5, True)
        with torch.no_grad():
            want = ops.graph_layernorm(x[:live].contiguous(), w, b, 1e-5, True)
        assert torch.equal(got[:live], want), live
    with pytest.raises(RuntimeError, match="forward-only"):
        with ops.live_rows(cnt):
            ops.graph_layernorm(x.clone().requires_grad_(True), w, b, 1e-5, True)
    with pytest.raises(ValueError):
        ops.live_rows(torch.tensor([3], device="cuda"))         # int64


@pytest.mark.parametrize("graph", [False, True])
@pytest.mark.parametrize("size", [5, 12])
def test_device_rollout_with_norm_model_equals_step_by_step_loop(graph, size):
    """A --norm=True model in the closed acting loop: the rollout's buffers are capacity-sized and their tail rows go stale as
    nodes are removed, so the whole-batch LayerNorm takes the live node total from the device (ops.live_rows).  It must play
    exactly the games the step-by-step API plays with exact-size batches (round 3 refused such models)."""
    from gnn_hex_amd.data import Batch
    from gnn_hex_amd.multi_env_manager import DeviceRollout, Env_manager
    hip, _ = _norm_pair(3, 35, seed=9, noisy=False)
    k, T = 6, 8
    a, b = Env_manager(k, size, gamma=0.9), Env_manager(k, size, gamma=0.9)
    a.reset(); b.reset()
    ro = DeviceRollout(a, hip, steps=T, eps=0.0, graph=graph)
    for _round in range(2):
        res = ro.run()
        obs = b.observe()
        for t in range(T):
            bt = Batch.from_data_list(obs)
            with torch.no_grad():
                adv = hip(bt.x, bt.edge_index, bt.batch, bt.ptr, advantages_only=True)
            vert, rank, _ = b.select_actions(adv, obs, eps=0.0)
            assert vert.cpu().tolist() == res.vertices[t].tolist(), "step %d" % t
            assert rank.cpu().tolist() == res.actions[t].tolist()
            assert obs.node_off == res.states[t].node_off
            obs, rew, dones, _ = b.step(vert)
            assert rew.tolist() == res.rewards[t].tolist() and dones.tolist() == res.dones[t].tolist()
        sa, sb = a._state(), b._state()
        for key in ("adj", "alive", "maker_turn", "total_moves"):
            assert (sa[key] == sb[key]).all(), key


def test_device_rollout_refuses_per_channel_cached_norm():
    """CachedGraphNorm (get_pre_defined("two_headed")) takes per-channel statistics over the batch's rows: still exact-size
    batches only."""
    from gnn_hex_amd.models import get_pre_defined
    from gnn_hex_amd.multi_env_manager import DeviceRollout, Env_manager
    args = Namespace(num_layers=3, hidden_channels=32, norm=True, noisy_dqn=False, noisy_sigma0=0.5, num_head_layers=2)
    model = get_pre_defined("two_headed", args).cuda()
    mgr = Env_manager(4, 5, device="cuda")
    mgr.reset()
    with pytest.raises(NotImplementedError, match="CachedGraphNorm"):
        DeviceRollout(mgr, model, steps=2, eps=0.0, graph=False)
