"""CPU: pins the model oracle (oracle/model_ref.py) with hand-derived known-answer tests and the structural
identities of the dueling head.  The oracle is PARITY-UNPINNED against the real
torch_geometric path (absent from the image); these tests are what pins it instead."""
import math

import numpy as np
import pytest
import torch

from helpers import batch_tensors, model_args
from oracle.model_ref import (SAGEConvRef, get_pre_defined_ref, scatter_ref)


def test_scatter_ref_kat():
    src = torch.tensor([[1., 5.], [3., 5.], [2., -1.], [7., 0.]])
    idx = torch.tensor([0, 0, 2, 2])
    assert torch.equal(scatter_ref(src, idx, reduce="sum"), torch.tensor([[4., 10.], [0., 0.], [9., -1.]]))
    assert torch.equal(scatter_ref(src, idx, reduce="mean"), torch.tensor([[2., 5.], [0., 0.], [4.5, -0.5]]))
    assert torch.equal(scatter_ref(src, idx, reduce="max"), torch.tensor([[3., 5.], [0., 0.], [7., 0.]]))
    assert torch.equal(scatter_ref(src, idx, reduce="min"), torch.tensor([[1., 5.], [0., 0.], [2., -1.]]))


def test_scatter_max_gradient_goes_to_first_index_on_ties():
    src = torch.tensor([[2.], [2.], [1.], [2.]], requires_grad=True)
    idx = torch.tensor([0, 0, 0, 0])
    scatter_ref(src, idx, reduce="max").sum().backward()
    assert torch.equal(src.grad, torch.tensor([[1.], [0.], [0.], [0.]]))
    src.grad = None
    scatter_ref(src, idx, reduce="min").sum().backward()
    assert torch.equal(src.grad, torch.tensor([[0.], [0.], [1.], [0.]]))


def test_sageconv_kat_constant_weights_and_isolated_node():
    # path 0-1-2 plus isolated node 3; all weights = c, bias = b  =>  y_i = c*sum(mean_nbrs) + b + c*sum(x_i)
    conv = SAGEConvRef(2, 3)
    c, b = 0.5, 0.25
    with torch.no_grad():
        conv.lin_l.weight.fill_(c); conv.lin_l.bias.fill_(b); conv.lin_r.weight.fill_(c)
    x = torch.tensor([[1., 2.], [3., 4.], [5., 6.], [7., 8.]])
    ei = torch.tensor([[0, 1, 1, 2], [1, 0, 2, 1]])
    y = conv(x, ei)
    mean = torch.tensor([[3., 4.], [3., 4.], [3., 4.], [0., 0.]])      # node 1: mean of nodes 0,2; node 3: isolated -> 0
    expect = (c * mean.sum(1) + b + c * x.sum(1)).view(-1, 1).expand(4, 3)
    assert torch.allclose(y, expect, atol=1e-6)


def test_hand_computed_tiny_network():
    """1 body layer + 1 head layer, hidden 2, a 3-node path graph: every number worked by hand in float64."""
    args = model_args(1, 2, head_layers=1)
    m = get_pre_defined_ref("modern_two_headed", args).double()
    with torch.no_grad():
        for p in m.parameters():
            p.fill_(0.1)
    x = torch.tensor([[1., 1., 1.], [2., 0., 1.], [1., 0., 1.]], dtype=torch.float64)
    ei = torch.tensor([[0, 1, 1, 2], [1, 0, 2, 1]])
    q = m(x, ei)
    w = 0.1
    X = x[:, :2].numpy()
    nb = {0: [1], 1: [0, 2], 2: [1]}

    def sage(F):
        out = np.zeros((3, 2))
        for i in range(3):
            mean = np.mean([F[j] for j in nb[i]], axis=0)
            out[i, :] = w * mean.sum() + w + w * F[i].sum()
        return np.maximum(out, 0)

    h = sage(sage(X))                      # body layer, then the (maker) head's gnn layer
    adv = 2 * np.tanh(h.sum(1) * w + w)
    pooled = np.concatenate([h.sum(0), h.max(0), h.min(0), h.mean(0)])
    z = max(pooled.sum() * w + w, 0.0)     # value MLP: Linear(8,1) -> relu -> Linear(1,1)
    v = math.tanh(z * w + w)
    expect = v + adv - adv.mean()
    assert np.allclose(q.detach().numpy(), expect, atol=1e-12)


@pytest.mark.parametrize("maker", [True, False])
def test_dueling_identities(maker):
    torch.manual_seed(0)
    m = get_pre_defined_ref("modern_two_headed", model_args(4, 12))
    x, ei, batch, ptr = batch_tensors("D1", [5, 6, 7], maker=maker)
    q = m(x, ei, batch, ptr)
    v, a = m(x, ei, batch, ptr, seperate=True)
    ao = m(x, ei, batch, ptr, advantages_only=True)
    assert q.shape == (x.shape[0],) and ao.shape == (x.shape[0], 1) and v.shape == (3,)
    assert q.abs().max() < 5
    for g in range(3):
        sl = slice(int(ptr[g]), int(ptr[g + 1]))
        assert abs(q[sl].mean().item() - v[g].item()) < 1e-6          # mean_g(Q) == tanh(value_g)
        assert abs(a[sl].mean().item()) < 1e-6
        assert torch.allclose(ao[sl, 0] - ao[sl, 0].mean(), a[sl], atol=1e-6)
    # the head is chosen by the side-to-move flag: the other head's parameters get no gradient
    q.sum().backward()
    used, unused = ("maker_head", "breaker_head") if maker else ("breaker_head", "maker_head")
    for k, p in m.named_parameters():
        if k.startswith(unused):
            assert p.grad is None
        elif k.startswith(used) or k.startswith("gnn"):
            assert p.grad is not None


def test_mixed_side_batch_is_rejected():
    m = get_pre_defined_ref("modern_two_headed", model_args(2, 4))
    x, ei, batch, ptr = batch_tensors("D0", [5, 5])
    x[3, 2] = 0.0
    with pytest.raises(AssertionError):
        m(x, ei, batch, ptr)


def test_parameter_counts_and_state_dict_keys():
    # GNN-S 37 382 parameters, GNN-L 486 974; key names of section 5 "Checkpoint"
    s = get_pre_defined_ref("modern_two_headed", model_args(10, 35))
    l = get_pre_defined_ref("modern_two_headed", model_args(15, 110))
    assert sum(p.numel() for p in s.parameters()) == 37382
    assert sum(p.numel() for p in l.parameters()) == 486974
    keys = set(s.state_dict().keys())
    for k in ("gnn.convs.0.lin_l.weight", "gnn.convs.9.lin_l.bias", "gnn.convs.3.lin_r.weight",
              "maker_head.gnn.convs.1.lin_l.weight", "breaker_head.linear.bias",
              "maker_head.value_head.layers.0.weight", "breaker_head.value_head.layers.1.bias"):
        assert k in keys
    assert "gnn.convs.0.lin_r.bias" not in keys
    assert s.state_dict()["gnn.convs.0.lin_l.weight"].shape == (35, 2)
    assert s.state_dict()["maker_head.value_head.layers.0.weight"].shape == (17, 140)


def _dense_numpy_forward(sd, layers, hidden, x, ei, batch, norm=False, noisy=False, eps=1e-5):
    """An INDEPENDENT float64 restatement of the network in dense linear algebra (numpy; adjacency-count matrices instead
    of gather / scatter, explicit loops instead of autograd modules).  It shares no code with oracle/model_ref.py: agreement
    between the two guards the oracle against restatement slips (it cannot pin either to torch_geometric)."""
    g = lambda k: sd[k].double().numpy()                                          # noqa: E731
    x = x.double().numpy()
    n = x.shape[0]
    src, dst = ei[0].numpy(), ei[1].numpy()
    A = np.zeros((n, n))
    np.add.at(A, (dst, src), 1.0)                    # A[i, j] = number of edges j -> i (duplicates counted)
    deg = np.maximum(A.sum(1, keepdims=True), 1.0)
    M = A / deg                                      # mean over in-edges; rows without in-edges stay zero
    maker = x[0, 2] == 1

    def ln(h, prefix):
        c = h - h.mean()
        return c / (c.std() + eps) * g(prefix + ".weight") + g(prefix + ".bias")

    def stack(h, prefix, count):
        for i in range(count):
            h = (M @ h) @ g("%s.convs.%d.lin_l.weight" % (prefix, i)).T + g("%s.convs.%d.lin_l.bias" % (prefix, i)) \
                + h @ g("%s.convs.%d.lin_r.weight" % (prefix, i)).T
            if norm:
                h = ln(h, "%s.norms.%d" % (prefix, i))
            h = np.maximum(h, 0.0)
        return h

    emb = stack(x[:, :2], "gnn", layers)
    if norm:
        emb = ln(emb, "after_embed_norm")
    head = "maker_head" if maker else "breaker_head"
    h = stack(emb, head + ".gnn", 2)
    if noisy:
        w = g(head + ".linear.weight_mu") + g(head + ".linear.weight_sigma") * g(head + ".linear.weight_epsilon")
        b = g(head + ".linear.bias_mu") + g(head + ".linear.bias_sigma") * g(head + ".linear.bias_epsilon")
    else:
        w, b = g(head + ".linear.weight"), g(head + ".linear.bias")
    adv = 2.0 * np.tanh(h @ w.T + b)[:, 0]
    bt = batch.numpy()
    q = np.zeros(n)
    for gi in range(int(bt.max()) + 1):
        rows = np.nonzero(bt == gi)[0]
        hg = h[rows]
        pooled = np.concatenate([hg.sum(0), hg.max(0), hg.min(0), hg.mean(0)])
        z = np.maximum(g(head + ".value_head.layers.0.weight") @ pooled + g(head + ".value_head.layers.0.bias"), 0.0)
        v = np.tanh(g(head + ".value_head.layers.1.weight") @ z + g(head + ".value_head.layers.1.bias"))[0]
        q[rows] = v + adv[rows] - adv[rows].mean()
    return q


@pytest.mark.parametrize("norm,noisy", [(False, False), (True, False), (False, True), (True, True)])
@pytest.mark.parametrize("maker", [True, False])
def test_oracle_agrees_with_an_independent_dense_restatement(norm, noisy, maker):
    from argparse import Namespace
    torch.manual_seed(5)
    args = Namespace(num_layers=4, hidden_channels=12, norm=norm, noisy_dqn=noisy, noisy_sigma0=0.5, num_head_layers=2)
    ref = get_pre_defined_ref("modern_two_headed", args).double()
    with torch.no_grad():
        for k, p in ref.named_parameters():
            if "norm" in k:
                p.add_(torch.randn(p.shape, dtype=torch.float64) * 0.3)
    # board graphs plus a directed / duplicated / isolated-node graph
    x, ei, batch, ptr = batch_tensors("D1", [5, 7, 6], maker=maker)
    n0 = x.shape[0]
    extra_x = torch.tensor([[3., 1., x[0, 2]], [0., 1., x[0, 2]], [2., 0., x[0, 2]], [1., 0., x[0, 2]]])
    extra_ei = torch.tensor([[0, 0, 2, 2], [1, 1, 1, 0]]) + n0          # duplicate edge 0->1, node 3 isolated
    x = torch.cat([x, extra_x]); ei = torch.cat([ei, extra_ei], 1)
    batch = torch.cat([batch, torch.full((4,), 3)])
    with torch.no_grad():
        q_ref = ref(x.double(), ei, batch).numpy()
    q_dense = _dense_numpy_forward(ref.state_dict(), 4, 12, x, ei, batch, norm=norm, noisy=noisy)
    assert np.abs(q_ref - q_dense).max() < 1e-10


# ---- the two_headed family (GN0/models.py:901-918): CachedGraphNorm + linear value head over mean pooling ------------------

def test_cached_graph_norm_known_answer_and_cache():
    """CachedGraphNormRef (GN0/models.py:644-670) against the formula worked in numpy: per-channel statistics over all nodes,
    mean_scale inside the centring, eps inside the square root; then the cache protocol of the reference."""
    from oracle.model_ref import CachedGraphNormRef
    rng = np.random.default_rng(3)
    x = rng.normal(size=(9, 4)) * 2.0 + 0.7
    norm = CachedGraphNormRef(4).double()
    with torch.no_grad():
        norm.weight.copy_(torch.tensor([1.5, 0.5, 2.0, 1.0]))
        norm.bias.copy_(torch.tensor([0.1, -0.2, 0.0, 0.3]))
        norm.mean_scale.copy_(torch.tensor([1.0, 0.5, 0.0, 1.2]))
    w, b, ms = (p.detach().numpy() for p in (norm.weight, norm.bias, norm.mean_scale))
    mean = x.mean(0)
    out = x - mean * ms
    var = (out ** 2).mean(0)
    want = w * out / np.sqrt(var + 1e-5) + b
    got = norm(torch.from_numpy(x)).detach().numpy()
    assert np.abs(got - want).max() < 1e-12
    # set_cache stores [1, C] statistics; use_cache on OTHER data normalises with them
    norm(torch.from_numpy(x), set_cache=True)
    assert norm.mean_cache.shape == (1, 4) and norm.var_cache.shape == (1, 4)
    assert np.abs(norm.mean_cache.numpy()[0] - mean).max() < 1e-12 and np.abs(norm.var_cache.detach().numpy()[0] - var).max() < 1e-12
    x2 = rng.normal(size=(5, 4))
    got2 = norm(torch.from_numpy(x2), use_cache=True).detach().numpy()
    want2 = w * (x2 - mean * ms) / np.sqrt(var + 1e-5) + b
    assert np.abs(got2 - want2).max() < 1e-12
    # set_cache wins over use_cache (GN0/models.py:656,663: `use_cache and not set_cache`)
    got3 = norm(torch.from_numpy(x2), set_cache=True, use_cache=True).detach().numpy()
    o3 = x2 - x2.mean(0) * ms
    assert np.abs(got3 - (w * o3 / np.sqrt((o3 ** 2).mean(0) + 1e-5) + b)).max() < 1e-12


def _dense_numpy_two_headed(sd, layers, x, ei, batch, norm, eps=1e-5):
    """Independent dense float64 restatement of get_pre_defined("two_headed") (shares no code with oracle/model_ref.py)."""
    g = lambda k: sd[k].double().numpy()                                          # noqa: E731
    x = x.double().numpy()
    n = x.shape[0]
    A = np.zeros((n, n))
    np.add.at(A, (ei[1].numpy(), ei[0].numpy()), 1.0)
    M = A / np.maximum(A.sum(1, keepdims=True), 1.0)
    maker = x[0, 2] == 1

    def gn(h, prefix):
        o = h - h.mean(0, keepdims=True) * g(prefix + ".mean_scale")
        return g(prefix + ".weight") * o / np.sqrt((o * o).mean(0, keepdims=True) + eps) + g(prefix + ".bias")

    def stack(h, prefix, count):
        for i in range(count):
            h = (M @ h) @ g("%s.convs.%d.lin_l.weight" % (prefix, i)).T + g("%s.convs.%d.lin_l.bias" % (prefix, i)) \
                + h @ g("%s.convs.%d.lin_r.weight" % (prefix, i)).T
            if norm:
                h = gn(h, "%s.norms.%d" % (prefix, i))
            h = np.maximum(h, 0.0)
        return h

    emb = stack(x[:, :2], "gnn", layers)
    if norm:
        emb = gn(emb, "after_embed_norm")
    head = "maker_head" if maker else "breaker_head"
    h = stack(emb, head + ".gnn", 2)
    adv = 2.0 * np.tanh(h @ g(head + ".linear.weight").T + g(head + ".linear.bias"))[:, 0]
    bt = batch.numpy()
    q = np.zeros(n)
    for gi in range(int(bt.max()) + 1):
        rows = np.nonzero(bt == gi)[0]
        v = np.tanh(g(head + ".value_head.weight") @ h[rows].mean(0) + g(head + ".value_head.bias"))[0]
        q[rows] = v + adv[rows] - adv[rows].mean()
    return q


@pytest.mark.parametrize("norm", [False, True])
@pytest.mark.parametrize("maker", [True, False])
def test_two_headed_oracle_agrees_with_an_independent_dense_restatement(norm, maker):
    from argparse import Namespace
    torch.manual_seed(9)
    args = Namespace(num_layers=3, hidden_channels=10, norm=norm, noisy_dqn=False, noisy_sigma0=0.5, num_head_layers=2)
    ref = get_pre_defined_ref("two_headed", args).double()
    with torch.no_grad():
        for k, p in ref.named_parameters():
            if "norm" in k:
                p.add_(torch.randn(p.shape, dtype=torch.float64) * 0.3)
    x, ei, batch, ptr = batch_tensors("D1", [5, 7, 6], maker=maker)
    with torch.no_grad():
        q_ref = ref(x.double(), ei, batch).numpy()
    assert np.abs(q_ref - _dense_numpy_two_headed(ref.state_dict(), 3, x, ei, batch, norm)).max() < 1e-10
    keys = list(ref.state_dict().keys())
    assert "maker_head.value_head.weight" in keys and ref.state_dict()["maker_head.value_head.weight"].shape == (1, 10)
    assert ("gnn.norms.0.mean_scale" in keys) == norm


def test_two_headed_oracle_norm_cache_protocol():
    """set_cache in eval mode stores the statistics of that batch in every norm of body and evaluated head; a later eval
    forward of ANOTHER batch uses them (different from its fresh-statistics result), training mode ignores them, and
    export_norm_cache / import_norm_cache move them to a second model (GN0/models.py:165-182, 513-535)."""
    from argparse import Namespace
    torch.manual_seed(11)
    args = Namespace(num_layers=2, hidden_channels=8, norm=True, noisy_dqn=False, noisy_sigma0=0.5, num_head_layers=2)
    ref = get_pre_defined_ref("two_headed", args).double().eval()
    xa, eia, ba, _ = batch_tensors("D1", [5, 6], maker=True)
    xb, eib, bb, _ = batch_tensors("D1", [7], maker=True)
    with torch.no_grad():
        fresh_b = ref(xb.double(), eib, bb)
        ref(xa.double(), eia, ba, set_cache=True)
        assert ref.gnn.has_cache and ref.maker_head.gnn.has_cache and not ref.breaker_head.gnn.has_cache
        cached_b = ref(xb.double(), eib, bb)
        assert (cached_b - fresh_b).abs().max() > 1e-6
        ref.train()
        assert torch.equal(ref(xb.double(), eib, bb), fresh_b)
        ref.eval()
        other = get_pre_defined_ref("two_headed", args).double().eval()
        other.load_state_dict(ref.state_dict())
        caches = [ref.gnn.export_norm_cache(), ref.maker_head.export_norm_cache(), None]
        assert caches[0][0].shape == (2, 1, 8)
        other.import_norm_cache(*caches)
        assert torch.equal(other(xb.double(), eib, bb), cached_b)
