"""The update of an UNMODIFIED training loop (torch indexing, torch's loss function, loss.backward()) on the fused loss path:
gnn_hex_amd/qvalues.py recognises the expression on the model's output tensor.  Reference loop: RainbowDQN agent behind
train.py (README.md:5,7: --loss_fn=mse, --prioritized_er=True).  Every form must give exactly the numbers of
ops.td_loss + ops.backward, and what is NOT recognised must stay plain torch with torch's result."""
import pytest
import torch
import torch.nn.functional as F

from helpers import batch_tensors, make_pair, sel_and_targets

pytestmark = pytest.mark.gpu


def _setup(b=24):
    dev = torch.device("cuda", 0)
    hip, ref = make_pair(10, 35, seed=4, device=dev)
    x, ei, bv, ptr = batch_tensors("D1", [7] * b, maker=True)
    sel, tgt = sel_and_targets(ptr)
    torch.manual_seed(2)
    w = torch.rand(b) + 0.5
    return hip, [t.to(dev) for t in (x, ei, bv, ptr, sel, tgt, w)]


def _grads(m):
    return {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}


def _reference(hip, d, weights, kind):
    from gnn_hex_amd import ops
    hip.zero_grad(set_to_none=True)
    q = hip(d[0], d[1], d[2], d[3])
    loss, td = ops.td_loss(q, d[4], d[5], d[6] if weights else None, kind)
    ops.backward(loss)
    return loss.detach().clone(), td.clone(), _grads(hip)


@pytest.mark.parametrize("select", ["getitem", "index_select", "gather", "take"])
@pytest.mark.parametrize("lossf", ["mse", "huber", "smooth_l1"])
def test_mean_reduction_is_the_fused_loss(select, lossf):
    from gnn_hex_amd import qvalues
    hip, d = _setup()
    kind = "mse" if lossf == "mse" else "huber"
    loss0, td0, g0 = _reference(hip, d, False, kind)
    hip.zero_grad(set_to_none=True)
    q = hip(d[0], d[1], d[2], d[3])
    assert type(q) is qvalues.QValues
    est = {"getitem": lambda: q[d[4]], "index_select": lambda: q.index_select(0, d[4]),
           "gather": lambda: torch.gather(q, 0, d[4]), "take": lambda: q.take(d[4])}[select]()
    assert type(est) is qvalues.QSelected
    loss = {"mse": F.mse_loss, "huber": F.huber_loss, "smooth_l1": F.smooth_l1_loss}[lossf](est, d[5])
    assert type(loss) is qvalues.TdLoss
    td_errors = est - d[5]                       # the priorities' input: plain torch on real values
    assert type(td_errors) is torch.Tensor and torch.equal(td_errors.detach(), td0)
    loss.backward()
    torch.cuda.synchronize()
    assert torch.equal(loss.detach(), loss0)
    g = _grads(hip)
    assert g.keys() == g0.keys()
    for k in g0:
        assert torch.equal(g[k], g0[k]), k


@pytest.mark.parametrize("order", ["w*l", "l*w"])
def test_prioritized_replay_form(order):
    """reduction='none', importance weights, mean -- the --prioritized_er=True form."""
    from gnn_hex_amd import qvalues
    hip, d = _setup()
    loss0, td0, g0 = _reference(hip, d, True, "mse")
    hip.zero_grad(set_to_none=True)
    q = hip(d[0], d[1], d[2], d[3])
    losses = torch.nn.MSELoss(reduction="none")(q[d[4]], d[5])
    assert type(losses) is qvalues.QLosses
    assert torch.allclose(losses.detach(), td0 * td0, rtol=1e-6, atol=0)        # real values underneath
    weighted = d[6] * losses if order == "w*l" else losses * d[6]
    loss = weighted.mean() if order == "w*l" else torch.mean(weighted)
    assert type(loss) is qvalues.TdLoss
    loss.backward()
    torch.cuda.synchronize()
    assert torch.equal(loss.detach(), loss0)
    g = _grads(hip)
    for k in g0:
        assert torch.equal(g[k], g0[k]), k


def test_unrecognised_forms_stay_plain_torch():
    from gnn_hex_amd import qvalues
    hip, d = _setup()
    loss0, td0, g0 = _reference(hip, d, False, "mse")
    # reduction='sum', an extra term, autograd's own entry point: torch's path, torch's numbers
    hip.zero_grad(set_to_none=True)
    q = hip(d[0], d[1], d[2], d[3])
    loss = F.mse_loss(q[d[4]], d[5], reduction="sum") / d[4].numel() + 0.0 * q.mean()
    assert type(loss) is torch.Tensor
    loss.backward()
    g = _grads(hip)
    for k in g0:
        assert (g[k] - g0[k]).abs().max().item() <= 2e-6 * max(1.0, g0[k].abs().max().item()), k
    # torch.autograd.backward on a recognised loss, and a scaled loss: autograd's path
    hip.zero_grad(set_to_none=True)
    q = hip(d[0], d[1], d[2], d[3])
    loss = F.mse_loss(q[d[4]], d[5])
    (loss * 2.0).backward()
    g = _grads(hip)
    for k in g0:
        assert (g[k] - 2.0 * g0[k]).abs().max().item() <= 4e-6 * max(1.0, g0[k].abs().max().item()), k
    # boolean masks / 2-D indices / slices are not selections of actions
    q = hip(d[0], d[1], d[2], d[3])
    assert type(q[3:9]) is torch.Tensor and type(q[q > 0]) is torch.Tensor and type(q.detach()) is torch.Tensor
    # switched off: ordinary tensors
    qvalues.set_enabled(False)
    try:
        assert type(hip(d[0], d[1], d[2], d[3])) is torch.Tensor
    finally:
        qvalues.set_enabled(True)
    with torch.no_grad():
        assert type(hip(d[0], d[1], d[2], d[3])) is torch.Tensor


def test_direct_gradient_guards():
    """(1) a head parameter updated in place between a forward and its backward is an error (the fused backward reads
    the head tail's weights live, and nothing else would notice); (2) parameters with post-accumulate-grad hooks
    (optimizer-in-backward) take the autograd form, where the hook fires."""
    from gnn_hex_amd import ops
    hip, d = _setup()
    q = hip(d[0], d[1], d[2], d[3])
    loss, _ = ops.td_loss(q, d[4], d[5])
    with torch.no_grad():
        hip.maker_head.linear.weight.add_(1e-3)
    with pytest.raises(RuntimeError, match="modified in place"):
        ops.backward(loss)
    # hook -> autograd form, the hook runs
    fired = []
    p0 = next(hip.gnn.parameters())
    h = p0.register_post_accumulate_grad_hook(lambda p: fired.append(p.grad.abs().sum().item()))
    try:
        hip.zero_grad(set_to_none=True)
        q = hip(d[0], d[1], d[2], d[3])
        F.mse_loss(q[d[4]], d[5]).backward()
        torch.cuda.synchronize()
        assert len(fired) == 1 and p0.grad is not None
    finally:
        h.remove()


def test_gradient_buffers_are_reused_only_when_nobody_holds_them():
    """The backward hands out the flat gradient buffer of an earlier step again once every view of it has been dropped
    (``zero_grad(set_to_none=True)``), and leaves it alone while ``p.grad`` or any other reference still points into it:
    gradients a caller keeps must keep their values, accumulation into uncleared gradients must add."""
    import torch.nn.functional as F
    from helpers import batch_tensors, make_pair, sel_and_targets
    from gnn_hex_amd import ops
    hip, _ = make_pair(3, 35, seed=2)
    x, ei, bv, ptr = batch_tensors("D0", [7] * 16)
    sel, tgt = sel_and_targets(ptr)
    xd = ops.attach_hints(x.cuda(), True, int((ptr[1:] - ptr[:-1]).max()))
    eid = ei.cuda()
    eid._hex_grouped = True
    bvd, ptrd, seld = bv.cuda(), ptr.cuda(), sel.cuda()
    plist = [p for p in hip.parameters()]

    def step(t, clear=True):
        if clear:
            hip.zero_grad(set_to_none=True)
        q = hip(xd, eid, bvd, ptrd)
        F.mse_loss(q[seld], t).backward()

    t1, t2 = tgt.cuda(), (tgt * -0.5 + 0.1).cuda()
    step(t1)
    g1 = [p.grad for p in plist if p.grad is not None]
    ptr1 = g1[0].data_ptr()
    want1 = [g.clone() for g in g1]
    step(t2)                                               # g1 is still referenced here: its buffer must not be written
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(g1, want1))
    g2 = [p.grad.clone() for p in plist if p.grad is not None]
    assert plist[0].grad.data_ptr() != ptr1
    del g1
    step(t1)                                               # nobody holds the first buffer any more: handed out again
    torch.cuda.synchronize()
    assert plist[0].grad.data_ptr() == ptr1
    assert all(torch.equal(p.grad, w) for p, w in zip([p for p in plist if p.grad is not None], want1))
    step(t2, clear=False)                                  # accumulation into gradients that are still there
    torch.cuda.synchronize()
    got = [p.grad for p in plist if p.grad is not None]
    assert all(torch.allclose(g, a + b, rtol=0, atol=1e-6) for g, a, b in zip(got, want1, g2))
