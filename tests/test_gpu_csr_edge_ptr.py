"""The grouped CSR build with the collation's own edge offsets (hexgnn_csr_build_grouped_pack_e): same CSR bit for bit as the
searching build, through the model; offsets that do not tile the edge list are flagged.  Reference collation:
Batch.from_data_list (torch_geometric 2.2.0; call site GN0/RainbowDQN/evaluate_elo.py:229)."""
import pytest
import torch

from helpers import batch_tensors, make_pair

pytestmark = pytest.mark.gpu


def _run(hip, x, ei, bv, ptr, edge_ptr):
    from gnn_hex_amd import ops
    dev = torch.device("cuda", 0)
    xd, eid, bvd, ptrd = (t.to(dev) for t in (x, ei, bv, ptr))
    ops.attach_hints(xd, True, int((ptr[1:] - ptr[:-1]).max()))
    eid._hex_grouped = True
    if edge_ptr is not None:
        eid._hex_edge_ptr = edge_ptr.to(dev)
    with torch.no_grad():
        q = hip(xd, eid, bvd, ptrd)
    call = hip.__dict__["_fca"]
    gs = call.gs
    torch.cuda.synchronize()
    return q.clone(), gs


def test_edge_offsets_give_the_same_csr_and_bad_offsets_are_flagged():
    dev = torch.device("cuda", 0)
    hip, _ = make_pair(10, 35, seed=0, device=dev)
    sizes = [5, 7, 6, 7, 5, 7, 7, 6] * 4
    x, ei, bv, ptr = batch_tensors("D1", sizes, maker=True)
    ecnt = torch.bincount(bv[ei[0]], minlength=len(sizes))
    eptr = torch.cat([torch.zeros(1, dtype=torch.long), ecnt.cumsum(0)])
    q0, g0 = _run(hip, x, ei, bv, ptr, None)
    q1, g1 = _run(hip, x, ei, bv, ptr, eptr)
    assert torch.equal(q0, q1)
    for name in ("rowptr", "col", "rowptr_t", "col_t", "invdeg"):
        assert torch.equal(getattr(g0, name), getattr(g1, name)), name
    g1.check()
    bad = eptr.clone()
    bad[3] += 2                      # graph 2's range swallows two edges of graph 3
    _, g2 = _run(hip, x, ei, bv, ptr, bad)
    with pytest.raises(IndexError):
        g2.check()
    bad = eptr.clone()
    bad[-1] -= 1                     # the ranges do not reach the end of the edge list
    _, g3 = _run(hip, x, ei, bv, ptr, bad)
    with pytest.raises(IndexError):
        g3.check()
