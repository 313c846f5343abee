"""CPU: pins the env oracle (oracle/env_ref.c) -- closed-form start graphs, the reference's CSR container
(oracle/_ref and the golden CSR made from it), the reference's winner-agreement property
(graph_game/test_graph_game.py:41-70, cpp_hex/hex_graph_game/tests/consistency_test.cpp:8-81), observation
invariants (GN0/util/convert_graph.py:77-122), n-step transition maths (multi_env_manager.py:113-165) and the
committed playout fixtures."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SIZES = {5: (27, 116), 6: (38, 174), 7: (51, 244), 8: (66, 326), 9: (83, 420), 10: (102, 526), 11: (123, 644),
         12: (146, 774), 13: (171, 916)}


def _csr_of(game):
    adj, alive = game.dump()
    nv = adj.shape[0]
    src, dst, es = [], [], [0]
    for v in range(nv):
        bits = np.unpackbits(adj[v].view(np.uint8), bitorder="little")[:nv]
        nb = np.nonzero(bits)[0]
        src += [v] * len(nb)
        dst += nb.tolist()
        es.append(len(src))
    return np.array(src, np.int32), np.array(dst, np.int32), np.array(es, np.int32)


@pytest.mark.parametrize("n", sorted(SIZES))
def test_start_graph_closed_form_and_golden_csr(hexref, n):
    g = hexref.RefGame(n)
    v, e = SIZES[n]
    assert g.num_vertices() == v == n * n + 2
    assert 2 * g.num_edges() == e == 2 * (2 * n + (n - 2) * (n - 1) + n * (n - 1) + (n - 1) ** 2)
    gold = np.load(os.path.join(GOLD, "start_graph_csr.npz"))
    s, t, es = _csr_of(g)
    assert np.array_equal(s, gold["s%d" % n]) and np.array_equal(t, gold["t%d" % n]) and np.array_equal(es, gold["es%d" % n])


def test_reference_graph_container_live(hexref):
    """When oracle/_ref is present: the reference's own Graph (graph.cpp) against the oracle's bit matrix under a
    random add/delete/clear sequence, and the start graph rebuilt live."""
    from oracle import ref_graph
    if not ref_graph.available():
        pytest.skip("oracle/_ref not built (reference tree absent)")
    s, t, es = ref_graph.start_graph_via_reference_container(11)
    gs, gt, ges = _csr_of(hexref.RefGame(11))
    assert np.array_equal(s, gs) and np.array_equal(t, gt) and np.array_equal(es, ges)
    rng = np.random.default_rng(0)
    nv = 40
    g = ref_graph.RefGraph(nv)
    adj = np.zeros((nv, nv), dtype=bool)
    for _ in range(3000):
        a, b = rng.integers(0, nv, 2)
        op = rng.integers(0, 10)
        if a == b:
            continue
        if op < 6:
            assert g.add_edge(int(a), int(b)) == (not adj[a, b])
            adj[a, b] = adj[b, a] = True
        elif op < 9:
            assert g.delete_edge(int(a), int(b)) == bool(adj[a, b])
            adj[a, b] = adj[b, a] = False
        else:
            g.clear_vertex(int(a))
            adj[a, :] = False
            adj[:, a] = False
        assert g.edge_exists(int(a), int(b)) == bool(adj[a, b])
    s, t, es = g.dump()
    rs, rt = np.nonzero(adj)                  # row-major => sorted adjacency rows, as the reference keeps them
    assert np.array_equal(s, rs) and np.array_equal(t, rt)
    assert np.array_equal(es, np.concatenate([[0], np.cumsum(adj.sum(1))]))


@pytest.mark.parametrize("size,games", [(5, 150), (7, 150), (11, 200)])
def test_winner_agreement_with_and_without_removal(hexref, size, games):
    rng = np.random.default_rng(size)
    for it in range(games):
        simple, fancy = hexref.RefGame(size), hexref.RefGame(size)
        while simple.who_won() is None:
            acts = simple.get_actions()
            mv = int(acts[rng.integers(len(acts))])
            simple.make_move(mv, False)
            if mv in set(fancy.get_actions().tolist()):
                if fancy.who_won() is None:
                    fancy.make_move(mv, True)
                else:
                    fancy.maker_turn = not fancy.maker_turn
            else:
                resp = fancy.get_response(mv, for_maker=not fancy.maker_turn)
                if resp is not None:
                    if resp in set(simple.get_actions().tolist()):
                        simple.make_move(resp, False)
                    else:
                        simple.maker_turn = not simple.maker_turn
                else:
                    fancy.maker_turn = not fancy.maker_turn
        assert simple.who_won() == fancy.who_won(), "game %d" % it


def test_observation_invariants(hexref):
    for g in range(20):
        game = hexref.random_position(7 + (g % 5), g, want_maker_turn=(g % 2 == 0))
        x, ei, bm = game.observe()
        n, e2 = x.shape[0], ei.shape[1]
        assert bm[0] == 0 and bm[1] == 1 and np.all(np.diff(bm) > 0)
        assert x[0, 1] == 1 and x[1, 1] == 1 and np.all(x[2:, 1] == 0)
        assert np.all(x[:, 2] == (1.0 if game.maker_turn else 0.0))
        deg = np.bincount(ei[1], minlength=n)
        assert np.array_equal(x[:, 0], deg.astype(np.float32))
        half = e2 // 2
        assert np.array_equal(ei[:, :half], ei[::-1, half:])          # second half = flipped first half
        assert np.array_equal(game.get_actions(), bm[2:])
        assert game.who_won() is None


def test_golden_playouts(hexref):
    from oracle.make_golden import state_digest
    gold = np.load(os.path.join(GOLD, "playouts.npz"))
    for gid in range(int(gold["num_games"][0])):
        size, maker_first, winner, nmoves = gold["g%d_meta" % gid].tolist()
        game = hexref.RefGame(size)
        game.maker_turn = bool(maker_first)
        dig = gold["g%d_digest" % gid]
        for i, mv in enumerate(gold["g%d_moves" % gid].tolist()):
            assert game.who_won() is None
            game.make_move(mv, remove_dead_and_captured=True)
            assert tuple(int(v) for v in dig[i]) == state_digest(game)
        assert {"m": 0, "b": 1}[game.who_won()] == winner and game.total_num_moves == nmoves


def test_step_reward_and_reset_semantics(hexref):
    """multi_env_manager.py:76-103: reward +1 for the mover that wins, finished envs restart with the side to move
    of the others, all envs keep one side to move."""
    mgr = hexref.RefEnvManager(6, 5, gamma=0.97)
    rng = np.random.default_rng(1)
    seen_done = 0
    for t in range(60):
        side = mgr.global_onturn
        acts = [int(a[rng.integers(len(a))]) for a in mgr.get_valid_actions()]
        obs, rew, done, infos = mgr.step(acts)
        assert mgr.global_onturn != side
        for i in range(6):
            assert (obs[i].x[0, 2] == 1.0) == (mgr.global_onturn == "m")
            if done[i]:
                seen_done += 1
                assert rew[i] in (1.0, -1.0)
                em = infos[i]["episode_metrics"]
                assert set(em) == {"return", "discounted_return", "length", "time"}
                assert abs(em["discounted_return"]) == pytest.approx(0.97 ** em["length"])
                assert obs[i].x.shape[0] == 27            # fresh Hex-5 start graph
            else:
                assert rew[i] == 0 and infos[i] == {}
    assert seen_done > 0


def test_get_transitions_hand_worked(hexref):
    """n-step, sign-alternating, gamma-discounted two-player returns (multi_env_manager.py:141-164)."""
    mgr = hexref.RefEnvManager(2, 5, gamma=0.5, n_steps=[1, 2])
    mk = lambda side: hexref.RefObs(np.array([[0, 1, side], [0, 1, side], [3, 0, side]], np.float32),
                                    np.zeros((2, 0), np.int64), np.array([0, 1, 2]))
    start = [mk(1), mk(1)]
    states = [[mk(0), mk(0)], [mk(1), mk(1)], [mk(0), mk(0)], [mk(1), mk(1)]]
    actions = [[2, 2]] * 4
    rewards = [[0.0, 0.0], [1.0, 0.0], [0.0, 0.0], [0.0, -1.0]]
    dones = [[False, False], [True, False], [False, False], [False, True]]
    expl = [[False, False]] * 4
    maker, breaker = mgr.get_transitions(start, states, actions, rewards, dones, expl)
    # i=0 (maker to move): n=1 needs len(sh)=5 > 2 ok. env0: r = r0 - r1 = -1, done at j=1 -> terminal transition.
    #                      env1: r = 0, next = sh[2][1], not done.   n=2: len(sh) > 4 ok.
    #                      env0: stops at j=1 (done) r=-1 ; env1: r = r0 - r1 + 0.5*(r2 - r3) = 0.5, done at j=3.
    m = [(t[2], t[4]) for t in maker]
    assert m[:4] == [(-1.0, True), (0.0, False), (-1.0, True), (0.5, True)]
    # terminal transitions point at the starting observation with the mover's side flag, without backmap
    assert maker[0][3].x.shape[0] == 27 and np.all(maker[0][3].x[:, 2] == 1) and not hasattr(maker[0][3], "backmap")
    assert not hasattr(maker[1][0], "backmap")
    # i=1 (breaker to move): n=1: env0 r = r1 - r2 = 1, done at j=1 ; env1 r = 0 not done
    b = [(t[2], t[4]) for t in breaker]
    assert b[:2] == [(1.0, True), (0.0, False)]
    # exploratory action after the first step prunes the longer transition
    expl2 = [[False, False], [False, True], [False, False], [False, False]]
    maker2, _ = mgr.get_transitions([mk(1), mk(1)], [[mk(0), mk(0)], [mk(1), mk(1)], [mk(0), mk(0)], [mk(1), mk(1)]],
                                    actions, rewards, dones, expl2)
    assert len(maker2) == len(maker) - 2     # env1 loses both its i=0 transitions (j=1 exploratory)
