"""The sub-modules of modern_two_headed are callable on their own, as in the reference: a bare SAGEConv.forward
(GN0/torch_script_models.py:52-73, no activation), the body CachifiedGNN.forward (GN0/models.py:261-294) and
HeadNetwork.forward (GN0/models.py:368-384: RAW advantages [N,1] and value [B,1]) -- values and gradients (inputs and
parameters) against the oracle's modules at 1e-4."""
import pytest
import torch

from helpers import batch_tensors, make_pair

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _grads_close(hip_mod, ref_mod, scale_floor=1.0):
    for (k, p), (_, pr) in zip(hip_mod.named_parameters(), ref_mod.named_parameters()):
        if pr.grad is None:
            assert p.grad is None, k
            continue
        assert p.grad is not None, k
        err = (p.grad.cpu() - pr.grad).abs().max().item()
        assert err < TOL * max(scale_floor, pr.grad.abs().max().item()), "%s grad err %g" % (k, err)


@pytest.mark.parametrize("hidden", [35, 110])
def test_head_network_forward_raw_outputs(hidden):
    hip, ref = make_pair(3, hidden, seed=31)
    x, ei, batch, ptr = batch_tensors("D1", [7, 5, 11, 6, 7])
    gen = torch.Generator().manual_seed(5)
    emb = torch.rand(x.shape[0], hidden, generator=gen)          # stands for the body embedding (post-ReLU: >= 0)
    for head_name in ("maker_head", "breaker_head"):
        hh, hr = getattr(hip, head_name), getattr(ref, head_name)
        er = emb.clone().requires_grad_(True)
        eh = emb.clone().cuda().requires_grad_(True)
        adv_r, val_r = hr(er, ei, batch)
        adv_h, val_h = hh(eh, ei.cuda(), batch.cuda())
        assert adv_h.shape == adv_r.shape == (x.shape[0], 1) and val_h.shape == val_r.shape == (5, 1)
        assert (adv_h.cpu() - adv_r).abs().max() < TOL and (val_h.cpu() - val_r).abs().max() < TOL
        wa = torch.randn(adv_r.shape, generator=gen)
        wv = torch.randn(val_r.shape, generator=gen)
        hr.zero_grad(set_to_none=True)
        hh.zero_grad(set_to_none=True)
        ((adv_r * wa).sum() + (val_r * wv).sum()).backward()
        ((adv_h * wa.cuda()).sum() + (val_h * wv.cuda()).sum()).backward()
        torch.cuda.synchronize()
        assert (eh.grad.cpu() - er.grad).abs().max() < TOL * max(1.0, er.grad.abs().max().item())
        _grads_close(hh, hr)
        # advantages_only: [N,1] raw advantages, no gradient for the value head
        hr.zero_grad(set_to_none=True)
        hh.zero_grad(set_to_none=True)
        ao_r = hr(emb, ei, batch, advantages_only=True)
        ao_h = hh(emb.cuda(), ei.cuda(), batch.cuda(), advantages_only=True)
        assert ao_h.shape == ao_r.shape and (ao_h.cpu() - ao_r).abs().max() < TOL
        ao_r.sum().backward()
        ao_h.sum().backward()
        _grads_close(hh, hr)


def test_sage_conv_forward_has_no_activation():
    hip, ref = make_pair(3, 48, seed=32)
    x, ei, batch, ptr = batch_tensors("D1", [7, 9, 5])
    gen = torch.Generator().manual_seed(6)
    # hidden -> hidden layer on signed inputs: the output must keep its negative entries
    xin = torch.randn(x.shape[0], 48, generator=gen)
    cr, ch = ref.gnn.convs[1], hip.gnn.convs[1]
    xr = xin.clone().requires_grad_(True)
    xh = xin.clone().cuda().requires_grad_(True)
    yr, yh = cr(xr, ei), ch(xh, ei.cuda())
    assert (yr < 0).any() and (yh.cpu() - yr).abs().max() < TOL
    w = torch.randn(yr.shape, generator=gen)
    cr.zero_grad(set_to_none=True); ch.zero_grad(set_to_none=True)
    (yr * w).sum().backward()
    (yh * w.cuda()).sum().backward()
    assert (xh.grad.cpu() - xr.grad).abs().max() < TOL * max(1.0, xr.grad.abs().max().item())
    _grads_close(ch, cr)
    # the raw first layer (2 -> hidden)
    c0r, c0h = ref.gnn.convs[0], hip.gnn.convs[0]
    y0r, y0h = c0r(x[:, :2], ei), c0h(x[:, :2].cuda(), ei.cuda())
    assert (y0r < 0).any() and (y0h.cpu() - y0r).abs().max() < TOL
    c0r.zero_grad(set_to_none=True); c0h.zero_grad(set_to_none=True)
    w0 = torch.randn(y0r.shape, generator=gen)
    (y0r * w0).sum().backward()
    (y0h * w0.cuda()).sum().backward()
    _grads_close(c0h, c0r)


def test_body_gnn_forward_matches():
    hip, ref = make_pair(4, 35, seed=33)
    x, ei, batch, ptr = batch_tensors("D1", [7] * 6)
    yr = ref.gnn(x[:, :2], ei)
    yh = hip.gnn(x[:, :2].cuda(), ei.cuda())
    assert (yh.cpu() - yr).abs().max() < TOL and (yh >= 0).all()
    yr.sum().backward()
    yh.sum().backward()
    _grads_close(hip.gnn, ref.gnn)
