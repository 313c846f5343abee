"""The HexAra head oracle (oracle/hexara_ref.py) against the reference's OWN known answers: rl_loop/unittest_model.py holds
exact output sizes, graph indices and batch pointers of SAGE_torch_script's swap / terminal-node surgery for three
hand-made batches (lines 16-52) and a randomized property check (lines 54-92: segment structure, pointer arithmetic,
sum(exp(pi)) == 1 per graph).  Those fixtures PIN the index logic; the float values stay "parity unpinned" (pyg absent)."""
import random

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


def random_case(rng, nprng):
    """rl_loop/unittest_model.py:57-80, same generator calls in the same order (python `random` seed 42, numpy seed 4)."""
    num_graphs = rng.randint(1, 30)
    num_nodes = rng.randint(3, 30) * num_graphs
    num_edges = rng.randint(30, 60) * num_graphs
    x = torch.zeros(num_nodes, 3)
    edge_index = torch.randint(0, num_nodes, (2, num_edges))
    gi = torch.tensor(list(sorted(sum([[i, i, i] for i in range(num_graphs)], [])
                                  + nprng.randint(0, num_graphs, num_nodes - 3 * num_graphs).tolist())), dtype=torch.long)
    bp = [0]
    cur = 0
    for i in range(len(gi)):
        if gi[i] != cur:
            bp.append(i)
            cur = gi[i]
    bp.append(len(gi))
    bp = torch.tensor(bp, dtype=torch.long)
    did_swap = []
    for start, fin in zip(bp[:-1], bp[1:]):
        if rng.random() > 0.5:
            did_swap.append(True)
            x[start:fin, 2] = 1
        else:
            did_swap.append(False)
    return x, edge_index, gi, bp, did_swap, num_graphs, num_nodes


def check_random_case(model_fn, case):
    x, ei, gi_in, bp_in, did_swap, num_graphs, num_nodes = case
    pi, value, gi, bp = model_fn(x, ei, gi_in, bp_in)
    assert value.shape == (num_graphs,) and bp.shape == (num_graphs + 1,)
    assert pi.shape == gi.shape == (num_nodes - num_graphs * 2 + int(np.sum(did_swap)),)
    minus = 0
    for i, (start, fin) in enumerate(zip(bp[:-1].tolist(), bp[1:].tolist())):
        assert bool((gi[start:fin] == gi[start]).all())
        if fin < len(gi):
            assert gi[start] != gi[fin]
        assert start == int(bp_in[i]) - minus
        assert abs(float(pi[start:fin].exp().sum()) - 1.0) < 1e-4
        minus += 2 - did_swap[i]


def test_reference_randomized_property():
    rng = random.Random(42)
    nprng = np.random.RandomState(4)
    torch.manual_seed(1)
    model = small_model(swap_allowed=True)
    for _ in range(25):
        case = random_case(rng, nprng)
        if int((case[3][1:] - case[3][:-1]).min()) <= 2:       # the model asserts > 2 nodes per graph (line 313)
            continue
        with torch.no_grad():
            check_random_case(model, case)


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
