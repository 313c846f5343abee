"""GPU parity: HIP path (through the C ABI) vs the CPU oracle on the same seeded inputs.

Tolerance: 1e-4 absolute on Q-values and on every parameter gradient (fp32; BASELINE.json north_star);
integer outputs (CSR, graph ptr) bit-exact.
"""
import numpy as np
import pytest
import torch

from helpers import batch_tensors, make_pair, sel_and_targets

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(params=[(True, "fp32"), (True, "bf16x3"), (False, "fp32")], ids=["fused", "fused-bf16x3", "layered"],
                autouse=True)
def _all_paths(request):
    """Every parity test runs through the fused per-graph kernels (exact fp32 MFMA and the split-precision bf16x3
    math) and through the general layer-major kernels (graphs > 128 nodes or hidden > 112 always take the latter).
    The SAME 1e-4 bar applies to all three."""
    from gnn_hex_amd import ops
    ops.set_fused(request.param[0])
    ops.set_math(request.param[1])
    yield
    ops.set_fused(True)
    ops.set_math("fp32")


def _step(model, x, ei, batch, ptr, sel, tgt, **kw):
    model.zero_grad(set_to_none=True)
    q = model(x, ei, batch, ptr, **kw)
    loss = torch.nn.functional.mse_loss(q[sel], tgt)
    loss.backward()
    return q.detach(), {k: (p.grad.detach().clone() if p.grad is not None else None)
                        for k, p in model.named_parameters()}


def _compare(hip, ref, x, ei, batch, ptr, tol=TOL):
    sel, tgt = sel_and_targets(ptr)
    q_ref, g_ref = _step(ref, x, ei, batch, ptr, sel, tgt)
    dev = "cuda"
    q_hip, g_hip = _step(hip, x.to(dev), ei.to(dev), batch.to(dev), ptr.to(dev), sel.to(dev), tgt.to(dev))
    torch.cuda.synchronize()
    assert q_hip.shape == q_ref.shape
    err = (q_hip.cpu() - q_ref).abs().max().item()
    assert err < tol, "Q max abs err %g" % err
    for k in g_ref:
        if g_ref[k] is None:
            assert g_hip[k] is None, "%s: reference grad is None, HIP grad is not" % k
            continue
        assert g_hip[k] is not None, "%s: missing grad" % k
        gerr = (g_hip[k].cpu() - g_ref[k]).abs().max().item()
        assert gerr < tol, "%s grad max abs err %g" % (k, gerr)
    return err


def test_csr_build_bit_exact():
    from gnn_hex_amd import ops
    x, ei, batch, ptr = batch_tensors("D1", [5, 7, 11, 6, 13, 9])
    n = x.shape[0]
    gs = ops.GraphStructure(ei.cuda(), n)
    torch.cuda.synchronize()
    src, dst = ei[0].numpy(), ei[1].numpy()
    order = np.lexsort((src, dst))
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(rowptr, dst + 1, 1)
    rowptr = np.cumsum(rowptr)
    assert np.array_equal(gs.rowptr.cpu().numpy(), rowptr)
    assert np.array_equal(gs.col.cpu().numpy()[: ei.shape[1]], src[order])
    order_t = np.lexsort((dst, src))
    rowptr_t = np.zeros(n + 1, dtype=np.int64)
    np.add.at(rowptr_t, src + 1, 1)
    assert np.array_equal(gs.rowptr_t.cpu().numpy(), np.cumsum(rowptr_t))
    assert np.array_equal(gs.col_t.cpu().numpy()[: ei.shape[1]], dst[order_t])
    deg = np.diff(rowptr)
    assert np.array_equal(gs.invdeg.cpu().numpy()[:n], (1.0 / np.maximum(deg, 1)).astype(np.float32))
    gs.check()
    gptr, b = ops.graph_ptr(batch.cuda(), None, n, "cuda")
    assert b == 6 and np.array_equal(gptr.cpu().numpy(), ptr.numpy())


@pytest.mark.parametrize("maker", [True, False])
def test_gnn_s_hex7_start_positions(maker):
    hip, ref = make_pair(10, 35)
    x, ei, batch, ptr = batch_tensors("D0", [7] * 32, maker=maker)
    _compare(hip, ref, x, ei, batch, ptr)


def test_gnn_s_hex7_random_positions():
    hip, ref = make_pair(10, 35, seed=3)
    x, ei, batch, ptr = batch_tensors("D1", [7] * 32)
    _compare(hip, ref, x, ei, batch, ptr)


def test_gnn_l_hex11():
    hip, ref = make_pair(15, 110)
    x, ei, batch, ptr = batch_tensors("D1", [11] * 16, maker=False)
    _compare(hip, ref, x, ei, batch, ptr)


def test_mixed_sizes_ragged():
    hip, ref = make_pair(15, 110, seed=5)
    sizes = [5 + (g % 9) for g in range(27)]
    x, ei, batch, ptr = batch_tensors("D1", sizes)
    _compare(hip, ref, x, ei, batch, ptr)


@pytest.mark.parametrize("hidden", [16, 33, 64, 128])
def test_other_widths(hidden):
    hip, ref = make_pair(3, hidden, seed=hidden)
    x, ei, batch, ptr = batch_tensors("D1", [5, 6, 7, 8])
    _compare(hip, ref, x, ei, batch, ptr)


def test_seperate_and_advantages_only():
    hip, ref = make_pair(4, 35, seed=2)
    x, ei, batch, ptr = batch_tensors("D1", [7] * 8)
    xc, eic, bc, pc = x.cuda(), ei.cuda(), batch.cuda(), ptr.cuda()
    v_ref, a_ref = ref(x, ei, batch, ptr, seperate=True)
    v_hip, a_hip = hip(xc, eic, bc, pc, seperate=True)
    assert (v_hip.cpu() - v_ref).abs().max() < TOL and (a_hip.cpu() - a_ref).abs().max() < TOL
    ao_ref = ref(x, ei, batch, ptr, advantages_only=True)
    ao_hip = hip(xc, eic, bc, pc, advantages_only=True)
    assert ao_hip.shape == ao_ref.shape == (x.shape[0], 1)
    assert (ao_hip.cpu() - ao_ref).abs().max() < TOL
    # gradients through the two-output form.  This synthetic loss sums over ALL nodes, so gradients reach |g| ~ 30
    # (the metric's mse step stays below ~2): the 1e-4 bar is applied relative to the tensor's scale when that exceeds 1.
    for m, args in ((ref, (x, ei, batch, ptr)), (hip, (xc, eic, bc, pc))):
        m.zero_grad(set_to_none=True)
        v, a = m(*args, seperate=True)
        (v.sum() * 0.5 + (a * a).sum()).backward()
    for (k, p), (_, pr) in zip(hip.named_parameters(), ref.named_parameters()):
        if pr.grad is None:
            assert p.grad is None
        else:
            assert (p.grad.cpu() - pr.grad).abs().max() < TOL * max(1.0, pr.grad.abs().max().item()), k
    # advantages_only: the value head gets no gradient
    for m, args in ((ref, (x, ei, batch, ptr)), (hip, (xc, eic, bc, pc))):
        m.zero_grad(set_to_none=True)
        m(*args, advantages_only=True).sum().backward()
    for (k, p), (_, pr) in zip(hip.named_parameters(), ref.named_parameters()):
        if pr.grad is None:
            assert p.grad is None, k
        else:
            assert (p.grad.cpu() - pr.grad).abs().max() < TOL * max(1.0, pr.grad.abs().max().item()), k


def test_single_graph_no_batch_vector_and_no_grad():
    hip, ref = make_pair(5, 35, seed=7)
    x, ei, _, _ = batch_tensors("D0", [7])
    with torch.no_grad():
        q_ref = ref(x, ei)
        q_hip = hip(x.cuda(), ei.cuda())
    assert (q_hip.cpu() - q_ref).abs().max() < TOL
    # dueling identity (GN0/models.py:571-584): mean over the graph of Q equals tanh(value)
    v, a = hip(x.cuda(), ei.cuda(), seperate=True)
    assert abs(q_hip.mean().item() - v.item()) < 1e-5
    assert q_hip.abs().max() < 5


def test_mixed_side_batch_asserts_like_reference():
    hip, _ = make_pair(3, 16)
    x, ei, batch, ptr = batch_tensors("D0", [5, 5])
    x[0, 2] = 0.0
    with pytest.raises(AssertionError):
        hip(x.cuda(), ei.cuda(), batch.cuda(), ptr.cuda())


def test_final_conv_acts_and_hook():
    hip, ref = make_pair(4, 35, seed=9)
    x, ei, batch, ptr = batch_tensors("D1", [7] * 4)
    q_ref = ref(x, ei, batch, ptr)
    q_ref.sum().backward()
    q = hip(x.cuda(), ei.cuda(), batch.cuda(), ptr.cuda())
    q.sum().backward()
    assert (hip.final_conv_acts.cpu() - ref.final_conv_acts).abs().max() < TOL
    assert (hip.final_conv_grads.cpu() - ref.final_conv_grads).abs().max() < TOL


def test_cpu_tensors_fail_loudly():
    from gnn_hex_amd._lib import HexGnnError
    hip, _ = make_pair(3, 16)
    x, ei, batch, ptr = batch_tensors("D0", [5])
    with pytest.raises(HexGnnError):
        hip(x, ei, batch, ptr)
