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
