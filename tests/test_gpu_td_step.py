"""ops.td_step: the DQN update's loss formed in the forward kernel's tail (hexgnn_qnet_forward_td / _backward_flat_td).

Reference step (README.md:5,7): Q = model(x, edge_index, batch); loss = loss_fn(Q[sel], target) with the
prioritized replay's importance weights; loss.backward().  The fused form must give the same numbers as the three calls (td,
d loss / d Q and every gradient bit-identical, the loss in hexgnn_td_loss_forward's reduction shape) and the oracle's in float64."""
import pytest
import torch

from helpers import batch_tensors, make_pair, sel_and_targets

pytestmark = pytest.mark.gpu


def _grads(m):
    return {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}


def _batch(cfg, data, b, maker=True):
    sizes = [11] * b if cfg == "L" else [7] * b
    x, ei, bv, ptr = batch_tensors(data, sizes, maker=maker)
    sel, tgt = sel_and_targets(ptr)
    return x, ei, bv, ptr, sel, tgt


@pytest.mark.parametrize("cfg,layers,hidden", [("S", 10, 35), ("L", 15, 110)])
@pytest.mark.parametrize("data", ["D0", "D1"])
@pytest.mark.parametrize("loss_fn", ["mse", "huber"])
def test_td_step_equals_the_three_calls_and_the_float64_oracle(cfg, layers, hidden, data, loss_fn):
    from gnn_hex_amd import ops
    dev = torch.device("cuda", 0)
    hip, ref = make_pair(layers, hidden, seed=3, device=dev)
    b = 24
    x, ei, bv, ptr, sel, tgt = _batch(cfg, data, b)
    torch.manual_seed(5)
    w = torch.rand(b) + 0.5
    xd, eid, bvd, ptrd, seld, tgtd, wd = (t.to(dev) for t in (x, ei, bv, ptr, sel, tgt, w))
    # three calls
    hip.zero_grad(set_to_none=True)
    q = hip(xd, eid, bvd, ptrd)
    loss3, td3 = ops.td_loss(q, seld, tgtd, wd, loss_fn)
    ops.backward(loss3)
    g3 = _grads(hip)
    # fused
    hip.zero_grad(set_to_none=True)
    loss1, td1, q1 = ops.td_step(hip, xd, eid, bvd, ptrd, sel=seld, target=tgtd, weights=wd, loss_fn=loss_fn)
    torch.cuda.synchronize()
    call = q1._hex_call
    assert call.td is not None, "the fused form did not run (fell back to the three calls)"
    g1 = _grads(hip)
    assert torch.equal(q1.detach(), q.detach())
    assert torch.equal(td1, td3)
    assert torch.equal(loss1, loss3.detach())
    assert g1.keys() == g3.keys()
    for k in g3:
        assert torch.equal(g1[k], g3[k]), k
    # oracle in float64
    ref64 = ref.double()
    q64 = ref64(x.double(), ei, bv, ptr)
    d = q64[sel] - tgt.double()
    l = d * d if loss_fn == "mse" else torch.where(d.abs() <= 1, 0.5 * d * d, d.abs() - 0.5)
    loss64 = (w.double() * l).mean()
    loss64.backward()
    assert abs(loss1.item() - loss64.item()) <= 1e-5 * max(1.0, abs(loss64.item()))
    assert (td1.cpu().double() - d.detach()).abs().max().item() <= 1e-5
    for (k, p) in ref64.named_parameters():
        if p.grad is None:
            continue
        err = (g1[k].cpu().double() - p.grad).abs().max().item()
        assert err <= 1e-4, (k, err)


def test_td_step_without_weights_and_full_batch_graph_replay():
    """B = 256 start positions through a captured step: replays give the eager result bit for bit."""
    from gnn_hex_amd import ops
    from gnn_hex_amd.graphs import GraphedStep
    dev = torch.device("cuda", 0)
    hip, _ = make_pair(10, 35, seed=0, device=dev)
    x, ei, bv, ptr, sel, tgt = _batch("S", "D0", 256)
    xd, eid, bvd, ptrd, seld, tgtd = (t.to(dev) for t in (x, ei, bv, ptr, sel, tgt))
    ops.attach_hints(xd, True, int((ptr[1:] - ptr[:-1]).max()))       # no host sync inside a capture
    eid._hex_grouped = True                                            # collated graph by graph: the one-launch CSR build + pack
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        hip.zero_grad(set_to_none=True)
        loss_e, td_e, _ = ops.td_step(hip, xd, eid, bvd, ptrd, sel=seld, target=tgtd)
        torch.cuda.synchronize()
        ge = _grads(hip)
        loss_e, td_e = loss_e.clone(), td_e.clone()
    plist = list(hip.parameters())

    def fn():
        for p in plist:
            p.grad = None
        out = ops.td_step(hip, xd, eid, bvd, ptrd, sel=seld, target=tgtd)
        return out[0], out[1]

    g = GraphedStep(fn, plist)
    for _ in range(3):
        loss_g, td_g = g.replay()
    torch.cuda.synchronize()
    assert torch.equal(loss_g, loss_e) and torch.equal(td_g, td_e)
    gg = _grads(hip)
    for k in ge:
        assert torch.equal(gg[k], ge[k]), k


def test_td_step_selection_outside_its_graph_is_poisoned_and_flagged():
    from gnn_hex_amd import ops
    dev = torch.device("cuda", 0)
    hip, _ = make_pair(10, 35, seed=0, device=dev)
    x, ei, bv, ptr, sel, tgt = _batch("S", "D0", 8)
    sel = sel.clone()
    sel[3] = sel[4]                      # graph 3's entry names a node of graph 4
    xd, eid, bvd, ptrd, seld, tgtd = (t.to(dev) for t in (x, ei, bv, ptr, sel, tgt))
    loss, td, q = ops.td_step(hip, xd, eid, bvd, ptrd, sel=seld, target=tgtd)
    torch.cuda.synchronize()
    assert torch.isnan(loss).item() and torch.isnan(td[3]).item() and not torch.isnan(td[4]).item()
    with pytest.raises(IndexError):
        q._hex_call.gs.check()


def test_td_step_falls_back_when_the_selection_is_not_one_per_graph():
    from gnn_hex_amd import ops
    dev = torch.device("cuda", 0)
    hip, _ = make_pair(10, 35, seed=0, device=dev)
    x, ei, bv, ptr, sel, tgt = _batch("S", "D0", 8)
    sel2 = torch.cat([sel, sel[:3]])
    tgt2 = torch.cat([tgt, tgt[:3]])
    xd, eid, bvd, ptrd, seld, tgtd = (t.to(dev) for t in (x, ei, bv, ptr, sel2, tgt2))
    hip.zero_grad(set_to_none=True)
    loss, td, q = ops.td_step(hip, xd, eid, bvd, ptrd, sel=seld, target=tgtd)
    g1 = _grads(hip)
    hip.zero_grad(set_to_none=True)
    q3 = hip(xd, eid, bvd, ptrd)
    loss3, td3 = ops.td_loss(q3, seld, tgtd)
    ops.backward(loss3)
    g3 = _grads(hip)
    assert td.numel() == 11 and torch.equal(td, td3) and torch.equal(loss.detach(), loss3.detach())
    for k in g3:
        assert torch.equal(g1[k], g3[k]), k
