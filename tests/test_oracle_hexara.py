"""The HexAra head oracle (oracle/hexara_ref.py) against the reference's OWN known answers: rl_loop/unittest_model.py holds
exact output sizes, graph indices and batch pointers of SAGE_torch_script's swap / terminal-node surgery for three
hand-made batches (lines 16-52) and a randomized property check (lines 54-92: segment structure, pointer arithmetic,
sum(exp(pi)) == 1 per graph).  Those fixtures PIN the index logic; the float values stay "parity unpinned" (pyg absent)."""
import numpy as np
import pytest
import torch

from oracle.hexara_ref import get_current_model_ref, scatter_log_softmax_ref

# (x rows with feature 2 set, n, edge_index, graph_indices, batch_ptr) -> (pi size, value size, gi, bp)
#   rl_loop/unittest_model.py:16-27 (test_no_swap), 29-40 (test_all_swap), 42-52 (test_some_swap)
FIXTURES = {
    "no_swap": (dict(n=10, ones_from=None, ei=[[0, 3, 4, 5, 6, 8], [1, 2, 3, 7, 9, 6]], gi=[0] * 5 + [1] * 5, bp=[0, 5, 10]),
                dict(pi=6, value=2, gi=[0, 0, 0, 1, 1, 1], bp=[0, 3, 6])),
    "all_swap": (dict(n=10, ones_from=0, ei=[[0, 3, 4, 5, 6, 8], [1, 2, 3, 7, 9, 6]], gi=[0] * 5 + [1] * 5, bp=[0, 5, 10]),
                 dict(pi=8, value=2, gi=[0, 0, 0, 0, 1, 1, 1, 1], bp=[0, 4, 8])),
    "some_swap": (dict(n=14, ones_from=5, ei=[[0, 3, 4, 5, 6, 8, 11, 13, 12], [1, 2, 3, 7, 9, 6, 13, 11, 10]],
                       gi=[0] * 5 + [1] * 5 + [2] * 4, bp=[0, 5, 10, 14]),
                  dict(pi=10, value=3, gi=[0, 0, 0, 1, 1, 1, 1, 2, 2, 2], bp=[0, 3, 7, 10])),
}


def fixture_inputs(name):
    f, want = FIXTURES[name]
    if f["ones_from"] == 0:
        x = torch.ones(f["n"], 3)                      # unittest_model.py:30: torch.ones([10,3])
    else:
        x = torch.zeros(f["n"], 3)
        if f["ones_from"] is not None:
            x[f["ones_from"]:, 2] = 1
    return x, torch.tensor(f["ei"]), torch.tensor(f["gi"]), torch.tensor(f["bp"]), want


def small_model(swap_allowed, seed=0):
    torch.manual_seed(seed)
    return get_current_model_ref(hidden_channels=12, hidden_layers=3, policy_layers=2, value_layers=2, swap_allowed=swap_allowed)


@pytest.mark.parametrize("name", ["no_swap", "all_swap", "some_swap"])
def test_reference_unittest_fixtures(name):
    x, ei, gi, bp, want = fixture_inputs(name)
    model = small_model(swap_allowed=True)
    with torch.no_grad():
        pi, value, ogi, obp = model(x, ei, gi, bp)
    assert pi.shape == (want["pi"],) and value.shape == (want["value"],)
    assert ogi.tolist() == want["gi"] and obp.tolist() == want["bp"]
    for s, e in zip(obp[:-1], obp[1:]):
        assert abs(float(pi[s:e].exp().sum()) - 1.0) < 1e-5


def test_swap_disallowed_only_drops_the_terminals():
    """swap_allowed=False (get_current_model's default, torch_script_models.py:495): unittest_model.py:16-27 still holds,
    and the all-ones batch keeps the no-swap shape."""
    model = small_model(swap_allowed=False)
    for name in ("no_swap", "all_swap"):
        x, ei, gi, bp, _ = fixture_inputs(name)
        with torch.no_grad():
            pi, value, ogi, obp = model(x, ei, gi, bp)
        assert pi.shape == (6,) and ogi.tolist() == [0, 0, 0, 1, 1, 1] and obp.tolist() == [0, 3, 6]


def random_case(gen):
    """A random batch in the spirit of the reference's randomized check (rl_loop/unittest_model.py:54-92: up to 30 graphs of
    3..30+ nodes, random multigraph edges, swapping possible in about half of the graphs) from this suite's own generator."""
    num_graphs = int(gen.integers(1, 31))
    sizes = gen.integers(3, 31, size=num_graphs)
    bp = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.long)
    num_nodes = int(bp[-1])
    gi = torch.repeat_interleave(torch.arange(num_graphs), torch.as_tensor(sizes))
    edge_index = torch.as_tensor(gen.integers(0, num_nodes, size=(2, int(gen.integers(30, 61)) * num_graphs)))
    did_swap = [bool(v) for v in gen.random(num_graphs) > 0.5]
    x = torch.zeros(num_nodes, 3)
    for g, flag in enumerate(did_swap):
        if flag:
            x[bp[g]:bp[g + 1], 2] = 1
    return x, edge_index, gi, bp, did_swap, num_graphs, num_nodes


def check_random_case(model_fn, case):
    """The invariants the reference asserts for every output (unittest_model.py:80-92): one value per graph; one policy entry
    per non-terminal node plus one per graph that may swap; every segment belongs to one graph and starts where the input
    graph started minus the entries dropped before it; the probabilities of a segment sum to one."""
    x, ei, gi_in, bp_in, did_swap, num_graphs, num_nodes = case
    pi, value, gi, bp = model_fn(x, ei, gi_in, bp_in)
    assert value.shape == (num_graphs,) and bp.shape == (num_graphs + 1,)
    assert pi.shape == gi.shape == (num_nodes - 2 * num_graphs + sum(did_swap),)
    dropped = 0
    for g in range(num_graphs):
        start, fin = int(bp[g]), int(bp[g + 1])
        assert start == int(bp_in[g]) - dropped
        assert fin - start == int(bp_in[g + 1] - bp_in[g]) - 2 + int(did_swap[g])
        assert bool((gi[start:fin] == g).all())
        assert abs(float(pi[start:fin].exp().sum()) - 1.0) < 1e-4
        dropped += 2 - int(did_swap[g])


def test_reference_randomized_property():
    gen = np.random.default_rng(42)
    torch.manual_seed(1)
    model = small_model(swap_allowed=True)
    for _ in range(25):
        with torch.no_grad():
            check_random_case(model, random_case(gen))


def test_scatter_log_softmax_known_answer():
    src = torch.tensor([1.0, 2.0, 3.0, -1.0, 0.5])
    idx = torch.tensor([0, 0, 0, 1, 1])
    want = torch.cat([torch.log_softmax(src[:3], 0), torch.log_softmax(src[3:], 0)])
    assert (scatter_log_softmax_ref(src, idx) - want).abs().max() < 1e-6


def test_state_dict_layout():
    """Key names of the reference module tree (torch_script_models.py:291-303): gnn.convs.*, my_modules.{value_head,
    policy_head}.convs.*, my_modules.{value_linear,swap_linear}.layers.*; the policy head ends in a SAGEConv(H, 1)."""
    m = small_model(True)
    keys = list(m.state_dict().keys())
    assert keys[0] == "gnn.convs.0.lin_l.weight" and "my_modules.policy_head.convs.1.lin_r.weight" in keys
    assert m.state_dict()["my_modules.policy_head.convs.1.lin_l.weight"].shape == (1, 12)
    assert m.state_dict()["my_modules.value_linear.layers.0.weight"].shape == (6, 48)
    assert m.state_dict()["my_modules.swap_linear.layers.1.weight"].shape == (1, 6)
