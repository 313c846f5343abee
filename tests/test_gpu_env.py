"""GPU parity of the HIP board-graph builder (through the C ABI) against the C env oracle: adjacency bit matrices,
alive sets, winners, move counts, response sets, observations (x / edge_index / backmap / CSR) -- all bit-exact --
plus the Env_manager step contract and the committed playout fixtures."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _oracle_state(game):
    adj, alive = game.dump()
    return adj, alive


def _assert_env_equals_oracle(st, i, game):
    adj, alive = game.dump()
    assert np.array_equal(st["alive"][i], alive), "alive set differs (env %d)" % i
    assert np.array_equal(st["adj"][i], adj), "adjacency differs (env %d)" % i
    assert bool(st["maker_turn"][i]) == game.maker_turn
    assert int(st["total_moves"][i]) == game.total_num_moves


@pytest.mark.parametrize("size,num_envs,steps", [(5, 16, 40), (7, 32, 60), (11, 64, 80), (13, 8, 60), (15, 4, 170), (17, 3, 220),
                                                 (19, 2, 300), (25, 2, 520), (6, 8, 40), (3, 5, 12)])
def test_lockstep_random_play_bit_exact(hexref, size, num_envs, steps):
    from gnn_hex_amd.multi_env_manager import Env_manager
    mgr = Env_manager(num_envs, size, gamma=0.97)
    ref = hexref.RefEnvManager(num_envs, size, gamma=0.97)
    obs = mgr.reset()
    robs = ref.reset()
    rng = np.random.default_rng(size)
    finished = 0
    for t in range(steps):
        valid = mgr.get_valid_actions()
        rvalid = ref.get_valid_actions()
        acts = []
        for i in range(num_envs):
            assert np.array_equal(valid[i], rvalid[i])
            acts.append(int(valid[i][rng.integers(len(valid[i]))]))
        obs, rew, done, infos = mgr.step(acts)
        robs, rrew, rdone, rinfos = ref.step(acts)
        assert np.array_equal(rew, rrew) and np.array_equal(done, rdone)
        assert mgr.global_onturn == ref.global_onturn
        st = mgr._state()
        for i in range(num_envs):
            _assert_env_equals_oracle(st, i, ref.envs[i])
            if done[i]:
                finished += 1
                for k in ("return", "discounted_return", "length"):
                    assert infos[i]["episode_metrics"][k] == rinfos[i]["episode_metrics"][k]
            # observation: x, edge_index, backmap bit-exact against convert_node_switching_game(old_style=True)
            d = obs[i]
            assert np.array_equal(d.x.cpu().numpy(), robs[i].x)
            assert np.array_equal(d.edge_index.cpu().numpy(), robs[i].edge_index)
            assert np.array_equal(d.backmap.cpu().numpy(), robs[i].backmap)
    assert finished > 0


@pytest.mark.parametrize("size,num_envs,steps", [(11, 128, 70), (7, 128, 40), (11, 1024, 45)])
def test_lockstep_at_benchmark_env_counts(hexref, size, num_envs, steps):
    """The env counts of BASELINE.json configs 2-4 (parallel_envs = 128 and 1024): every env's adjacency bit matrix,
    alive set, side and move count bit-exact against the oracle after every step; rewards / dones / episode metrics
    equal; the BATCHED observation (x, edge_index, backmap, ptr) bit-exact against the collated oracle observations."""
    from gnn_hex_amd.data import Batch
    from gnn_hex_amd.multi_env_manager import Env_manager
    mgr = Env_manager(num_envs, size, gamma=0.97)
    ref = hexref.RefEnvManager(num_envs, size, gamma=0.97)
    mgr.reset()
    ref.reset()
    rng = np.random.default_rng(1000 * size + num_envs)
    finished = 0
    for t in range(steps):
        rvalid = ref.get_valid_actions()
        acts = [int(v[rng.integers(len(v))]) for v in rvalid]
        obs, rew, done, infos = mgr.step(acts)
        robs, rrew, rdone, rinfos = ref.step(acts)
        assert np.array_equal(rew, rrew) and np.array_equal(done, rdone)
        assert mgr.global_onturn == ref.global_onturn
        st = mgr._state()
        for i in range(num_envs):
            _assert_env_equals_oracle(st, i, ref.envs[i])
            if done[i]:
                finished += 1
                for k in ("return", "discounted_return", "length"):
                    assert infos[i]["episode_metrics"][k] == rinfos[i]["episode_metrics"][k]
        b = Batch.from_data_list(obs)
        offs = np.cumsum([0] + [o.x.shape[0] for o in robs])
        assert np.array_equal(b.ptr.cpu().numpy(), offs)
        assert np.array_equal(b.x.cpu().numpy(), np.concatenate([o.x for o in robs], 0))
        assert np.array_equal(b.edge_index.cpu().numpy(),
                              np.concatenate([o.edge_index + int(off) for o, off in zip(robs, offs[:-1])], 1))
        assert np.array_equal(obs.backmap.cpu().numpy(), np.concatenate([o.backmap for o in robs]))
    assert finished >= num_envs // 4


def test_batched_observation_matches_collation_and_csr(hexref):
    from gnn_hex_amd import ops
    from gnn_hex_amd.data import Batch
    from gnn_hex_amd.multi_env_manager import Env_manager
    mgr = Env_manager(24, 7)
    rng = np.random.default_rng(0)
    obs = mgr.reset()
    for _ in range(9):
        acts = [int(v[rng.integers(len(v))]) for v in mgr.get_valid_actions()]
        obs, _, _, _ = mgr.step(acts)
    b = Batch.from_data_list(obs)
    # equals a plain collation of the per-env Data objects
    xs = torch.cat([d.x for d in obs], 0)
    offs = np.cumsum([0] + [d.x.shape[0] for d in obs])
    eis = torch.cat([d.edge_index + int(o) for d, o in zip(obs, offs[:-1])], 1)
    assert torch.equal(b.x, xs) and torch.equal(b.edge_index, eis)
    assert np.array_equal(b.ptr.cpu().numpy(), offs)
    assert torch.equal(b.batch, torch.repeat_interleave(torch.arange(24), torch.tensor(np.diff(offs))).cuda())
    # the CSR emitted by the builder equals the CSR built from edge_index
    gs = b.edge_index._hex_csr
    gs2 = ops.GraphStructure(b.edge_index, b.x.shape[0])
    torch.cuda.synchronize()
    assert torch.equal(gs.rowptr, gs2.rowptr) and torch.equal(gs.col[:gs.e], gs2.col[:gs2.e])
    assert torch.equal(gs.rowptr, gs2.rowptr_t) and torch.equal(gs.col[:gs.e], gs2.col_t[:gs2.e])
    assert torch.equal(gs.invdeg[:gs.n], gs2.invdeg[:gs2.n])
    assert b.x._hex_is_maker == (mgr.global_onturn == "m")


def test_golden_playouts_on_device(hexref):
    """The committed fixtures (tests/golden/playouts.npz) replayed on the GPU, one env per game."""
    from oracle.make_golden import fnv1a64
    from gnn_hex_amd import _lib, ops
    import ctypes as C
    gold = np.load(os.path.join(GOLD, "playouts.npz"))
    L = _lib.lib()
    for gid in range(int(gold["num_games"][0])):
        size, maker_first, winner, nmoves = gold["g%d_meta" % gid].tolist()
        h = C.c_void_p()
        _lib.check(L.hexgnn_env_create(1, size, C.byref(h)))
        nv, words = L.hexgnn_env_num_vertices(h), L.hexgnn_env_words(h)
        _lib.check(L.hexgnn_env_set_maker_turn(h, int(maker_first), ops._stream()))
        res = torch.empty((1, 5), dtype=torch.int32, device="cuda")
        adj = torch.empty((1, nv, words), dtype=torch.int64, device="cuda")
        alive = torch.empty((1, nv), dtype=torch.uint8, device="cuda")
        mt = torch.empty(1, dtype=torch.int32, device="cuda")
        tm = torch.empty(1, dtype=torch.int32, device="cuda")
        dig = gold["g%d_digest" % gid]
        moves = gold["g%d_moves" % gid].tolist()
        for i, mv in enumerate(moves):
            act = torch.tensor([mv], dtype=torch.int32, device="cuda")
            _lib.check(L.hexgnn_env_step(h, act.data_ptr(), 1, 0, 0, res.data_ptr(), ops._stream()))
            _lib.check(L.hexgnn_env_export(h, adj.data_ptr(), alive.data_ptr(), mt.data_ptr(), tm.data_ptr(), None, None,
                                           ops._stream()))
            r = res.cpu().numpy()[0]
            assert r[4] == 0
            assert (r[0] >= 0) == (i == len(moves) - 1)
            if r[0] < 0:    # the residual graph of a DECIDED game is unspecified (include/hexgnn.h); only winner/length are
                a, al = adj.cpu().numpy().view(np.uint64)[0], alive.cpu().numpy()[0]
                assert (int(al.sum()), int(r[3]) // 2, fnv1a64(a.tobytes() + al.tobytes())) == tuple(int(v) for v in dig[i])
        assert int(r[0]) == winner and int(r[1]) == nmoves
        L.hexgnn_env_destroy(h)


def test_illegal_action_is_reported():
    from gnn_hex_amd.multi_env_manager import Env_manager
    mgr = Env_manager(4, 5)
    mgr.reset()
    with pytest.raises(ValueError):
        mgr.step([2, 3, 0, 4])        # vertex 0 is a terminal


def test_env_to_model_closed_loop(hexref):
    """Env observation -> Batch -> Q-network (CSR straight from the builder) -> greedy action -> env step, checked
    against the CPU oracle model on the same observation."""
    from helpers import make_pair
    from gnn_hex_amd.data import Batch
    from gnn_hex_amd.multi_env_manager import Env_manager
    hip, ref = make_pair(6, 35, seed=11)
    mgr = Env_manager(16, 7)
    obs = mgr.reset()
    for t in range(6):
        b = Batch.from_data_list(obs)
        with torch.no_grad():
            q = hip.simple_forward(b)
            q_ref = ref(b.x.cpu(), b.edge_index.cpu(), b.batch.cpu(), b.ptr.cpu())
        assert (q.cpu() - q_ref).abs().max() < 1e-4
        ptr = b.ptr.tolist()
        acts = [int(torch.argmax(q[ptr[g] + 2:ptr[g + 1]])) + 2 for g in range(16)]   # evaluate_elo.py:253-266
        obs, rew, done, infos = mgr.step(mgr.validate_actions(obs, acts))


def test_select_actions_greedy_and_exploratory():
    """hexgnn_select_actions vs the reference's per-graph torch.argmax over values[ptr[g]+2 : ptr[g+1]] + backmap."""
    from gnn_hex_amd.multi_env_manager import Env_manager
    mgr = Env_manager(20, 7)
    obs = mgr.reset()
    rng = np.random.default_rng(2)
    for _ in range(5):
        obs, *_ = mgr.step([int(v[rng.integers(len(v))]) for v in mgr.get_valid_actions()])
    gen = torch.Generator(device="cuda").manual_seed(0)
    q = torch.randn(obs.x.shape[0], device="cuda", generator=gen)
    q[obs.node_off[3] + 5] = q[obs.node_off[3] + 9] = 100.0            # a tie: the first index wins
    vert, rank, expl = mgr.select_actions(q, obs, eps=0.0)
    bm = obs.backmap.cpu()
    for g in range(20):
        n0, n1 = obs.node_off[g], obs.node_off[g + 1]
        want = int(torch.argmax(q[n0 + 2:n1])) + 2
        assert int(rank[g]) == want and int(vert[g]) == int(bm[n0 + want])
    assert int(rank[3]) == 5 and not expl.any()
    assert vert.tolist() == mgr.validate_actions(obs, rank.tolist())
    # eps = 1: every env explores, uniformly over its legal nodes, and the move is legal
    vert, rank, expl = mgr.select_actions(q, obs, eps=1.0, generator=gen)
    assert expl.all()
    for g in range(20):
        assert 2 <= int(rank[g]) < obs.node_off[g + 1] - obs.node_off[g]
    obs2, rew, done, infos = mgr.step(vert)                              # device actions go straight into the step
    assert len(obs2) == 20


@pytest.mark.parametrize("graph", [False, True])
@pytest.mark.parametrize("size", [5, 12], ids=["hex5", "hex12-layer-major"])
def test_device_rollout_equals_step_by_step_loop(graph, size):
    """DeviceRollout (observation -> Q forward -> greedy action -> step, sizes kept on the device, optionally one HIP
    graph) must play exactly the games the public step-by-step API plays, and leave the envs in the same state."""
    import torch
    from gnn_hex_amd.data import Batch
    from gnn_hex_amd.multi_env_manager import DeviceRollout, Env_manager
    from helpers import make_pair
    hip, _ = make_pair(3, 35, seed=4)
    # (Hex-12 boards have 146 nodes: the model runs on the layer-major kernels over the rollout's capacity-sized buffers,
    # whose rows past the shrinking total must read as empty)
    k, T = 8, 12
    a, b = Env_manager(k, size, gamma=0.9), Env_manager(k, size, gamma=0.9)
    a.reset(); b.reset()
    ro = DeviceRollout(a, hip, steps=T, eps=0.0, graph=graph)
    for _round in range(2):          # the second run starts from mid-game boards and replays the captured graph
        res = ro.run()
        obs = b.observe()
        for t in range(T):
            bt = Batch.from_data_list(obs)
            with torch.no_grad():
                adv = hip(bt.x, bt.edge_index, bt.batch, bt.ptr, advantages_only=True)
            vert, rank, _ = b.select_actions(adv, obs, eps=0.0)
            assert vert.cpu().tolist() == res.vertices[t].tolist(), "step %d" % t
            assert rank.cpu().tolist() == res.actions[t].tolist()
            assert obs.node_off == res.states[t].node_off and obs.edge_off == res.states[t].edge_off
            assert torch.equal(obs.snapshot()[0], res.states[t].snapshot()[0])
            obs, rew, dones, infos = b.step(vert)
            assert rew.tolist() == res.rewards[t].tolist() and dones.tolist() == res.dones[t].tolist()
            for i in range(k):
                assert ("episode_metrics" in infos[i]) == ("episode_metrics" in res.infos[t][i])
        if size == 5:
            assert res.dones.any(), "the rollout should finish at least one Hex-5 game"
        sa, sb = a._state(), b._state()
        for key in ("adj", "alive", "maker_turn", "total_moves"):
            assert (sa[key] == sb[key]).all(), key
        assert a.global_onturn == b.global_onturn and (a._sizes == b._sizes).all()


def test_split_rollout_equals_run_with_work_issued_in_between():
    """run_begin / run_end with GPU work issued between the two (a learner preparing its update while the actor plays)
    give the games run() gives; a second run_begin or a foreign handle is refused."""
    import torch
    from gnn_hex_amd.multi_env_manager import DeviceRollout, Env_manager
    from helpers import make_pair
    hip, _ = make_pair(3, 35, seed=4)
    a, b = Env_manager(8, 5, gamma=0.9), Env_manager(8, 5, gamma=0.9)
    a.reset(); b.reset()
    ra = DeviceRollout(a, hip, steps=8, eps=0.0, graph=True)
    rb = DeviceRollout(b, hip, steps=8, eps=0.0, graph=True)
    for _round in range(2):
        want = ra.run()
        h = rb.run_begin()
        with pytest.raises(RuntimeError, match="twice"):
            rb.run_begin()
        junk = torch.randn(512, 512, device="cuda") @ torch.randn(512, 512, device="cuda")   # queued behind the rollout
        got = rb.run_end(h)
        assert junk.isfinite().all()
        assert got.vertices.tolist() == want.vertices.tolist() and got.dones.tolist() == want.dones.tolist()
        assert got.rewards.tolist() == want.rewards.tolist()
        for sa, sb in zip(want.states, got.states):
            assert sa.node_off == sb.node_off and torch.equal(sa.snapshot()[0], sb.snapshot()[0])
    with pytest.raises(RuntimeError, match="handle"):
        rb.run_end(h)


def test_device_rollout_feeds_replay():
    """Rollout histories -> vectorised n-step assembly -> device replay ring -> sampled batches the model accepts."""
    import torch
    from gnn_hex_amd.multi_env_manager import DeviceRollout, Env_manager
    from gnn_hex_amd.replay import GraphReplayBuffer
    from helpers import make_pair
    hip, _ = make_pair(3, 35, seed=4)
    mgr = Env_manager(16, 5, gamma=0.97, n_steps=[2])
    mgr.reset()
    res = DeviceRollout(mgr, hip, steps=16, eps=0.3, graph=False).run()
    mb, bb = mgr.assemble_transitions(res.states[0], res.states[1:], list(res.actions), list(res.rewards),
                                      list(res.dones), list(res.exploratories))
    assert len(mb) > 0 and len(bb) > 0
    buf = GraphReplayBuffer(1024, 5, prioritized=True, alpha=0.5)
    buf.put_block(mb)
    idx, w, s, s2, act, r, d = buf.sample(32, beta=0.6)
    q = hip(s.x, s.edge_index, s.batch, s.ptr)
    assert q.shape[0] == s.x.shape[0] and torch.isfinite(q).all()
    assert int(act.max()) < int((s.ptr[1:] - s.ptr[:-1]).max())


def test_rollout_result_goes_stale_unless_detached():
    """RolloutResult.states view the rollout's snapshot ring: after the next run() they must refuse to be used (they would
    silently describe other boards); detach() gives them their own copies."""
    import torch
    from gnn_hex_amd.multi_env_manager import DeviceRollout, Env_manager
    from helpers import make_pair
    hip, _ = make_pair(3, 35, seed=4)
    mgr = Env_manager(8, 5, gamma=0.9)
    mgr.reset()
    ro = DeviceRollout(mgr, hip, steps=4, eps=0.5, graph=False)
    first = ro.run()
    kept = [tuple(t.clone() for t in s.snapshot()) for s in first.states]
    second = ro.run().detach()
    with pytest.raises(RuntimeError, match="stale rollout observation"):
        first.states[1].snapshot()
    copies = [tuple(t.clone() for t in s.snapshot()) for s in second.states]
    ro.run()
    for s, c in zip(second.states, copies):                      # detached: unaffected by the third run
        assert all(torch.equal(a, b) for a, b in zip(s.snapshot(), c))
    assert any(not torch.equal(k[0], c[0]) for k, c in zip(kept, copies))


def test_rollout_stitcher_covers_every_move_once():
    """Transitions across rollout boundaries: stitched assembly of three consecutive rollouts (each alone loses its last
    2 * n_step - 1 moves) must equal, as a multiset of (start step, env, action, reward, next step, done), the assembly of
    the concatenated histories in one call."""
    import torch
    from gnn_hex_amd.multi_env_manager import DeviceRollout, Env_manager, RolloutStitcher
    from helpers import make_pair
    hip, _ = make_pair(3, 35, seed=4)
    n_step, T, k = 2, 8, 12
    mgr = Env_manager(k, 5, gamma=0.97, n_steps=[n_step, 1])
    mgr.reset()
    ro = DeviceRollout(mgr, hip, steps=T, eps=0.4, graph=False)
    stitch = RolloutStitcher(mgr)
    runs, got = [], {True: [], False: []}
    for r in range(3):
        res = ro.run().detach()                       # detached: the one-shot assembly below reads all three again
        runs.append(res)
        off = max(0, r * T - stitch.keep)             # absolute move index of the stitched histories' first move
        for side, blk in zip((True, False), stitch.assemble(res)):
            for s_, e_, a_, rw, nx, d_ in zip(blk.src_step.tolist(), blk.env.tolist(), blk.action.tolist(),
                                              blk.reward.tolist(), blk.next_step.tolist(), blk.done.tolist()):
                got[side].append((s_ + off, e_, a_, round(rw, 9), -1 if nx < 0 else nx + off, d_))
    states = [runs[0].states[0]] + [s for res in runs for s in res.states[1:]]
    cat = lambda name: [x for res in runs for x in list(getattr(res, name))]       # noqa: E731
    blocks = mgr.assemble_transitions(states[0], states[1:], cat("actions"), cat("rewards"), cat("dones"),
                                      cat("exploratories"))
    for side, blk in zip((True, False), blocks):
        want = sorted(zip(blk.src_step.tolist(), blk.env.tolist(), blk.action.tolist(),
                          [round(v, 9) for v in blk.reward.tolist()],
                          [-1 if v < 0 else v for v in blk.next_step.tolist()], blk.done.tolist()))
        assert sorted(got[side]) == want and len(want) > 0
    # every move 0 .. 3T - 2*n_step - 1 of every env starts a 2-step transition unless pruned: more than a lone rollout gives
    lone = sum(len(b) for b in mgr.assemble_transitions(runs[0].states[0], runs[0].states[1:], list(runs[0].actions),
                                                        list(runs[0].rewards), list(runs[0].dones),
                                                        list(runs[0].exploratories)))
    assert sum(len(v) for v in got.values()) > 3 * lone


@pytest.mark.parametrize("total,world", [(128, 8), (1024, 8)])
def test_env_shards_equal_one_manager(total, world):
    """BASELINE config 4 shards parallel_envs=1024 eight ways (rank r owns envs [r*P/W, (r+1)*P/W), gnn_hex_amd.dist.
    shard_range).  Rehearsed on one GPU: W shard managers stepped with the slices of one action vector must stay identical,
    env for env, to ONE manager of all P envs -- states, rewards, dones, observations -- so a rank's shard is exactly its
    slice of the single-process run."""
    import torch
    from gnn_hex_amd.data import Batch
    from gnn_hex_amd.dist import shard_range
    from gnn_hex_amd.multi_env_manager import Env_manager
    size, steps = 11, 24
    whole = Env_manager(total, size, gamma=0.97)
    whole.reset()
    shards = []
    for r in range(world):
        lo, hi = shard_range(total, r, world)
        m = Env_manager(hi - lo, size, gamma=0.97)
        m.reset()
        shards.append((lo, hi, m))
    rng = np.random.default_rng(total)
    for t in range(steps):
        valid = whole.get_valid_actions()
        acts = [int(v[rng.integers(len(v))]) for v in valid]
        obs, rew, done, _ = whole.step(acts)
        st = whole._state()
        b = Batch.from_data_list(obs)
        for lo, hi, m in shards:
            o2, r2, d2, _ = m.step(acts[lo:hi])
            assert np.array_equal(r2, rew[lo:hi]) and np.array_equal(d2, done[lo:hi])
            s2 = m._state()
            for key in ("adj", "alive", "maker_turn", "total_moves"):
                assert np.array_equal(np.asarray(s2[key]), np.asarray(st[key])[lo:hi]), (key, lo)
            n0, n1 = obs.node_off[lo], obs.node_off[hi]
            assert torch.equal(o2.x, obs.x[n0:n1]) and torch.equal(o2.backmap, obs.backmap[n0:n1])
        assert b.x.shape[0] == obs.node_off[-1]
