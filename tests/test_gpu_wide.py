"""hidden_channels 129..256 (gnn_hex_amd/csrc/wide.hip): grow_width widens a model to any width (GN0/models.py:187-238,
497-508); beyond the 128 columns the LDS-resident kernels are compiled for the same entry points run plain kernels (mean
gather, exact-fp32 MFMA GEMM from L2, one wave per weight-gradient tile, simple head tails).  Q and every gradient against
the oracle, board graphs and graphs above 128 nodes, all three output modes, the stand-alone modules, a width curriculum
across the 128 boundary."""
import pytest
import torch

from helpers import batch_tensors, make_pair, sel_and_targets, sharpen_
from test_gpu_model import _compare, _random_batch

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.mark.parametrize("hidden,layers", [(144, 4), (200, 3), (256, 3), (130, 2)])
@pytest.mark.parametrize("maker", [True, False])
def test_wide_model_matches_oracle(hidden, layers, maker):
    hip, ref = make_pair(layers, hidden, seed=hidden)
    x, ei, batch, ptr = batch_tensors("D1", [7, 5, 11, 9, 13, 6], maker=maker)
    _compare(hip, ref, x, ei, batch, ptr)
    with torch.no_grad():
        dev = [t.cuda() for t in (x, ei, batch, ptr)]
        v, a = hip(*dev, seperate=True)
        v_ref, a_ref = ref(x, ei, batch, ptr, seperate=True)
        assert (v.cpu() - v_ref).abs().max().item() < TOL and (a.cpu() - a_ref).abs().max().item() < TOL
        adv = hip(*dev, advantages_only=True)
        assert (adv.cpu() - ref(x, ei, batch, ptr, advantages_only=True)).abs().max().item() < TOL
    assert (hip.final_conv_acts.cpu() - ref.final_conv_acts).abs().max().item() < TOL


def test_wide_model_sharp_weights_relative_gate():
    """A weight state whose signal does not collapse (helpers.sharpen_), gradients held norm-relative: an absolute 1e-4 gate
    cannot see an error on tensors whose largest entry is 1e-3."""
    hip, ref = make_pair(6, 160, seed=4)
    sharpen_(ref)
    hip.load_state_dict(ref.state_dict())
    x, ei, batch, ptr = batch_tensors("D1", [11] * 12, maker=True)
    _compare(hip, ref, x, ei, batch, ptr, grad_norm_rel=2e-4)


def test_wide_random_graphs_long_rows_and_final_conv_grads():
    hip, ref = make_pair(3, 176, seed=9)
    x, ei, batch, ptr = _random_batch([300, 40, 129, 7], seed=3, directed=True, p_edge=0.02)
    _compare(hip, ref, x, ei, batch, ptr)
    sel, tgt = sel_and_targets(ptr)
    for m, dev in ((ref, "cpu"), (hip, "cuda")):
        m.zero_grad(set_to_none=True)
        q = m(x.to(dev), ei.to(dev), batch.to(dev), ptr.to(dev))
        torch.nn.functional.mse_loss(q.reshape(-1)[sel.to(dev)], tgt.to(dev)).backward()
    g, g_ref = hip.final_conv_grads.cpu(), ref.final_conv_grads
    assert (g - g_ref).abs().max().item() < TOL * max(1.0, g_ref.abs().max().item())


def test_width_curriculum_across_128():
    """grow_width 110 -> 160 -> 256 on the device model: the widened model computes what the widened oracle weights give
    (zero-padded old weights: GN0/models.py:199-223) and trains on."""
    from gnn_hex_amd.models import get_pre_defined
    from oracle.model_ref import get_pre_defined_ref
    from helpers import model_args
    hip, ref = make_pair(3, 110, seed=2)
    x, ei, batch, ptr = batch_tensors("D1", [7, 9, 11], maker=True)
    dev = [t.cuda() for t in (x, ei, batch, ptr)]
    for width in (160, 256):
        hip.grow_width(width)
        assert hip.gnn.convs[1].lin_l.weight.shape == (width, width)
        ref2 = get_pre_defined_ref("modern_two_headed", model_args(3, width))
        ref2.load_state_dict(hip.state_dict())
        with torch.no_grad():
            q = hip(*dev)
            assert (q.cpu() - ref2(x, ei, batch, ptr)).abs().max().item() < TOL
        q = hip(*dev)
        hip.zero_grad(set_to_none=True)
        q.sum().backward()
        assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for k, p in hip.named_parameters()
                   if "breaker_head" not in k)


def test_wide_standalone_modules():
    """SAGEConv.forward and HeadNetwork.forward (raw outputs) at hidden 160."""
    hip, ref = make_pair(2, 160, seed=6)
    x, ei, batch, ptr = batch_tensors("D1", [7, 11], maker=True)
    h = torch.randn(x.shape[0], 160)
    with torch.no_grad():
        y = hip.gnn.convs[1](h.cuda(), ei.cuda())
        assert (y.cpu() - ref.gnn.convs[1](h, ei)).abs().max().item() < TOL
        adv, val = hip.maker_head(h.cuda(), ei.cuda(), batch.cuda())
        adv_r, val_r = ref.maker_head(h, ei, batch)
        assert (adv.cpu() - adv_r).abs().max().item() < TOL and (val.cpu() - val_r).abs().max().item() < TOL


def test_beyond_256_is_refused():
    from gnn_hex_amd._lib import HexGnnError
    hip, _ = make_pair(2, 272, seed=1)
    x, ei, batch, ptr = batch_tensors("D1", [5], maker=True)
    with pytest.raises(HexGnnError):
        hip(x.cuda(), ei.cuda(), batch.cuda(), ptr.cuda())
