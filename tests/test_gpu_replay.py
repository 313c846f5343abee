"""GPU: device-resident replay -- tree maintenance and stratified sampling bit-exact against oracle/replay_ref.py,
sampled batches identical to collating the stored observations, end-to-end with Env_manager.get_transitions."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_per_trees_and_sampling_match_oracle():
    from gnn_hex_amd import _lib, ops
    from oracle.replay_ref import SegmentTreePER
    L = _lib.lib()
    cap = 64
    st = torch.empty(2 * cap, dtype=torch.float64, device="cuda")
    mt = torch.empty(2 * cap, dtype=torch.float64, device="cuda")
    _lib.check(L.hexgnn_per_init(cap, st.data_ptr(), mt.data_ptr(), ops._stream()))
    ref = SegmentTreePER(cap)
    rng = np.random.default_rng(0)
    size = 0
    for rnd in range(6):
        k = 17
        idx = rng.choice(50, k, replace=False).astype(np.int32)
        pa = rng.random(k) ** 0.5 + 1e-3
        ref.update(idx, pa)
        size = max(size, int(idx.max()) + 1)
        di, dp = torch.from_numpy(idx).cuda(), torch.from_numpy(pa).cuda()
        _lib.check(L.hexgnn_per_update(cap, k, di.data_ptr(), dp.data_ptr(), st.data_ptr(), mt.data_ptr(), ops._stream()))
        torch.cuda.synchronize()
        assert np.array_equal(st.cpu().numpy()[1:], ref.sum[1:])          # fp64 sums in the same order: bit-exact
        fin = np.isfinite(ref.min)
        assert np.array_equal(mt.cpu().numpy()[fin], ref.min[fin])
    # every leaf < size must hold a priority for the weights to be defined: fill the gaps
    missing = np.array([i for i in range(size) if ref.sum[cap + i] == 0], dtype=np.int32)
    if len(missing):
        pa = np.full(len(missing), 0.25)
        ref.update(missing, pa)
        dm, dpa = torch.from_numpy(missing).cuda(), torch.from_numpy(pa).cuda()     # keep alive until the launch
        _lib.check(L.hexgnn_per_update(cap, len(missing), dm.data_ptr(), dpa.data_ptr(), st.data_ptr(), mt.data_ptr(),
                                       ops._stream()))
        torch.cuda.synchronize()
    for b, beta in ((32, 0.6), (7, 0.4), (64, 1.0)):
        u = rng.random(b)
        ri, rw = ref.sample(u, size, beta)
        oi = torch.empty(b, dtype=torch.int32, device="cuda")
        ow = torch.empty(b, dtype=torch.float32, device="cuda")
        du = torch.from_numpy(u).cuda()
        _lib.check(L.hexgnn_per_sample(cap, size, b, beta, du.data_ptr(), st.data_ptr(), mt.data_ptr(), oi.data_ptr(),
                                       ow.data_ptr(), ops._stream()))
        assert np.array_equal(oi.cpu().numpy(), ri)
        assert np.allclose(ow.cpu().numpy(), rw, rtol=1e-6, atol=0)


def test_per_trees_at_the_reference_buffer_size():
    """README.md:5,7: --buffer_size=260000 --batch_size=256 --prioritized_er_beta0=0.6.  Trees of capacity 2^18 filled
    in blocks of 2048 / 5000 entries (the launch-chunked update path), then 256-sample stratified draws: tree sums,
    sampled slots and importance weights against the oracle, bit for bit (fp64 sums in the same order)."""
    from gnn_hex_amd import _lib, ops
    from oracle.replay_ref import SegmentTreePER
    L = _lib.lib()
    cap = 1 << 18
    st = torch.empty(2 * cap, dtype=torch.float64, device="cuda")
    mt = torch.empty(2 * cap, dtype=torch.float64, device="cuda")
    _lib.check(L.hexgnn_per_init(cap, st.data_ptr(), mt.data_ptr(), ops._stream()))
    ref = SegmentTreePER(cap)
    rng = np.random.default_rng(11)
    size = 0
    for k in (2048, 5000, 2048):
        idx = np.arange(size, size + k, dtype=np.int32)              # ring slots filled in order
        pa = rng.random(k) ** 0.5 + 1e-3
        ref.update(idx, pa)
        di, dp = torch.from_numpy(idx).cuda(), torch.from_numpy(pa).cuda()
        _lib.check(L.hexgnn_per_update(cap, k, di.data_ptr(), dp.data_ptr(), st.data_ptr(), mt.data_ptr(), ops._stream()))
        size += k
    # a priority update of a sampled batch (with the duplicates sampling with replacement produces)
    upd = rng.integers(0, size, 256).astype(np.int32)
    upd[7] = upd[200]
    pa = rng.random(256) + 1e-3
    ref.update(upd, pa)
    du, dp = torch.from_numpy(upd).cuda(), torch.from_numpy(pa).cuda()
    _lib.check(L.hexgnn_per_update(cap, 256, du.data_ptr(), dp.data_ptr(), st.data_ptr(), mt.data_ptr(), ops._stream()))
    torch.cuda.synchronize()
    got_sum, got_min = st.cpu().numpy(), mt.cpu().numpy()
    assert np.array_equal(got_sum[1:], ref.sum[1:])
    fin = np.isfinite(ref.min)
    assert np.array_equal(got_min[fin], ref.min[fin])
    for beta in (0.6, 1.0):
        u = rng.random(256)
        ri, rw = ref.sample(u, size, beta)
        oi = torch.empty(256, dtype=torch.int32, device="cuda")
        ow = torch.empty(256, dtype=torch.float32, device="cuda")
        dv = torch.from_numpy(u).cuda()
        _lib.check(L.hexgnn_per_sample(cap, size, 256, beta, dv.data_ptr(), st.data_ptr(), mt.data_ptr(), oi.data_ptr(),
                                       ow.data_ptr(), ops._stream()))
        assert np.array_equal(oi.cpu().numpy(), ri)
        assert np.allclose(ow.cpu().numpy(), rw, rtol=1e-6, atol=0)


@pytest.mark.parametrize("bits", [32, 64])
def test_fused_priority_update_matches_oracle(bits):
    """hexgnn_per_update_td: leaf = (|td| + eps)^alpha with the last occurrence of a slot winning, running maximum raised to
    the largest priority; td == NULL stores (running maximum)^alpha -- against the oracle fed with numpy's values (pow may
    differ in the last place between libms: 1e-14 relative), over a list longer than one launch chunk."""
    from gnn_hex_amd import _lib, ops
    from oracle.replay_ref import SegmentTreePER
    L = _lib.lib()
    cap, alpha, eps = 1 << 13, 0.5, 1e-6
    st = torch.empty(2 * cap, dtype=torch.float64, device="cuda")
    mt = torch.empty(2 * cap, dtype=torch.float64, device="cuda")
    _lib.check(L.hexgnn_per_init(cap, st.data_ptr(), mt.data_ptr(), ops._stream()))
    mp = torch.ones((), dtype=torch.float64, device="cuda")
    ref = SegmentTreePER(cap)
    rng = np.random.default_rng(21)
    dt = np.int32 if bits == 32 else np.int64
    # 1. a block of new transitions at the running maximum (1.0)
    new = np.arange(0, 5000).astype(dt)
    ref.update(new.astype(np.int32), np.full(len(new), 1.0 ** alpha))
    dn = torch.from_numpy(new).cuda()
    _lib.check(L.hexgnn_per_update_td(cap, len(new), dn.data_ptr(), bits, None, alpha, eps, mp.data_ptr(), st.data_ptr(),
                                      mt.data_ptr(), ops._stream()))
    # 2. TD errors for a sampled batch with duplicates, longer than one chunk; one out-of-range slot is ignored
    k = 3000
    idx = rng.integers(0, 5000, k).astype(dt)
    idx[17] = idx[2900]
    idx[5] = cap + 3 if bits == 64 else -1
    td = (rng.standard_normal(k) * 3).astype(np.float32)
    p = np.abs(td).astype(np.float64) + eps
    keep = (idx >= 0) & (idx < cap)
    ref.update(idx[keep].astype(np.int32), p[keep] ** alpha)
    di, dtd = torch.from_numpy(idx).cuda(), torch.from_numpy(td).cuda()
    _lib.check(L.hexgnn_per_update_td(cap, k, di.data_ptr(), bits, dtd.data_ptr(), alpha, eps, mp.data_ptr(), st.data_ptr(),
                                      mt.data_ptr(), ops._stream()))
    torch.cuda.synchronize()
    assert float(mp) == max(1.0, float(p.max()))         # every entry counts for the maximum, as in the torch expression
    got_sum, got_min = st.cpu().numpy(), mt.cpu().numpy()
    assert np.allclose(got_sum[1:], ref.sum[1:], rtol=1e-13, atol=0)
    fin = np.isfinite(ref.min)
    assert np.allclose(got_min[fin], ref.min[fin], rtol=1e-13, atol=0) and np.array_equal(np.isfinite(got_min), fin)
    # 3. the next block of new transitions enters at the raised maximum
    new2 = np.arange(5000, 5100).astype(dt)
    dn2 = torch.from_numpy(new2).cuda()
    _lib.check(L.hexgnn_per_update_td(cap, len(new2), dn2.data_ptr(), bits, None, alpha, eps, mp.data_ptr(), st.data_ptr(),
                                      mt.data_ptr(), ops._stream()))
    torch.cuda.synchronize()
    assert np.allclose(st.cpu().numpy()[cap + 5000:cap + 5100], float(mp) ** alpha, rtol=1e-14, atol=0)


def test_per_update_duplicates_last_wins_and_large_lists():
    """PER samples with replacement, so update_priorities sees duplicated slots: the last occurrence must win in BOTH
    trees (oracle: sequential loop), deterministically, also across the 2048-entry launch chunks of a long list."""
    from gnn_hex_amd import _lib, ops
    from oracle.replay_ref import SegmentTreePER
    L = _lib.lib()
    cap = 4096
    rng = np.random.default_rng(7)
    for k in (300, 2048, 5000):
        st = torch.empty(2 * cap, dtype=torch.float64, device="cuda")
        mt = torch.empty(2 * cap, dtype=torch.float64, device="cuda")
        _lib.check(L.hexgnn_per_init(cap, st.data_ptr(), mt.data_ptr(), ops._stream()))
        ref = SegmentTreePER(cap)
        idx = rng.integers(0, 700, k).astype(np.int32)          # heavy duplication
        idx[-1] = idx[0]                                         # a duplicate spanning the whole list
        pa = rng.random(k) + 1e-3
        ref.update(idx, pa)
        di, dp = torch.from_numpy(idx).cuda(), torch.from_numpy(pa).cuda()
        for _ in range(3):                                       # repeated: no run-to-run variation
            _lib.check(L.hexgnn_per_update(cap, k, di.data_ptr(), dp.data_ptr(), st.data_ptr(), mt.data_ptr(), ops._stream()))
            torch.cuda.synchronize()
            assert np.array_equal(st.cpu().numpy()[1:], ref.sum[1:])
            fin = np.isfinite(ref.min)
            assert np.array_equal(mt.cpu().numpy()[fin], ref.min[fin])
            leaves = np.unique(idx)
            assert np.array_equal(st.cpu().numpy()[cap + leaves], mt.cpu().numpy()[cap + leaves])


def _play(mgr, steps, rng):
    obs0 = mgr.reset()
    states, actions, rewards, dones, expl = [], [], [], [], []
    obs = obs0
    for _ in range(steps):
        acts_rank = [int(rng.integers(2, obs.node_off[i + 1] - obs.node_off[i])) for i in range(mgr.num_envs)]
        obs2, r, d, _ = mgr.step(mgr.validate_actions(obs, acts_rank))
        states.append(obs2); actions.append(acts_rank); rewards.append(r); dones.append(d)
        expl.append(np.zeros(mgr.num_envs, dtype=bool))
        obs = obs2
    return obs0, states, actions, rewards, dones, expl


def test_replay_round_trip_with_env_transitions():
    from gnn_hex_amd.data import Batch
    from gnn_hex_amd.multi_env_manager import Env_manager
    from gnn_hex_amd.replay import GraphReplayBuffer
    rng = np.random.default_rng(3)
    mgr = Env_manager(12, 5, gamma=0.97, n_steps=[2])
    obs0, states, actions, rewards, dones, expl = _play(mgr, 14, rng)
    # keep plain copies of what the transitions look like before the buffer sees them
    maker, breaker = mgr.get_transitions(obs0, states, actions, rewards, dones, expl)
    assert len(maker) > 10 and len(breaker) > 10
    buf = GraphReplayBuffer(64, 5, prioritized=True, alpha=0.5)
    buf.put(maker)
    assert len(buf) == min(64, len(maker))
    kept = maker[-64:] if len(maker) > 64 else maker
    gen = torch.Generator(device="cuda").manual_seed(0)
    idx, w, s, s2, a, r, d = buf.sample(16, beta=0.6, generator=gen)
    assert torch.all(w == 1.0)                      # all priorities equal so far
    ih = idx.cpu().tolist()
    ref_s = Batch.from_data_list([kept[i][0] for i in ih])
    ref_s2 = Batch.from_data_list([kept[i][3] for i in ih])
    for got, want in ((s, ref_s), (s2, ref_s2)):
        assert torch.equal(got.x, want.x) and torch.equal(got.edge_index, want.edge_index)
        assert torch.equal(got.ptr, want.ptr) and torch.equal(got.batch, want.batch)
    assert a.cpu().tolist() == [int(kept[i][1]) for i in ih]
    assert np.allclose(r.cpu().numpy(), [kept[i][2] for i in ih])
    assert d.cpu().tolist() == [bool(kept[i][4]) for i in ih]
    assert s.x._hex_is_maker is True and s2.x._hex_is_maker is True
    # terminal transitions point at the start position
    for i in ih:
        if kept[i][4]:
            j = ih.index(i)
            assert int(s2.ptr[j + 1] - s2.ptr[j]) == 27
    # priorities: |td| = 50 -> p^alpha = sqrt(50 + eps) = 7.07 against 63 entries at 1: drawn with probability
    # 7.07 / 70.07 = 10 %, importance weight (p/p_min)^-beta = 7.07^-0.6 = 0.309
    buf.update_priorities(idx[:1], torch.tensor([50.0], device="cuda"))
    idx2, w2, *_ = buf.sample(64, beta=0.6, generator=gen)
    top = int(idx[0])
    frac = (idx2 == top).float().mean().item()
    assert 0.07 < frac < 0.15
    assert abs(float(w2[idx2 == top][0]) - 50.000001 ** (-0.3)) < 1e-4 and float(w2[idx2 != top][0]) == 1.0
    # wrap-around keeps the newest transitions
    buf.put(breaker)
    assert len(buf) == 64


def test_split_sampling_equals_sample_and_guards_against_stores():
    """sample_begin / sample_end (the draw started early, the batch built later: examples/selfplay_train.py) return what
    sample() returns for the same generator state -- also with another buffer's draw and a priority update of that other
    buffer in between -- and refuse to finish a draw across a put()."""
    from gnn_hex_amd.multi_env_manager import Env_manager
    from gnn_hex_amd.replay import GraphReplayBuffer
    rng = np.random.default_rng(11)
    mgr = Env_manager(12, 5, gamma=0.97, n_steps=[2])
    obs0, states, actions, rewards, dones, expl = _play(mgr, 14, rng)
    maker, breaker = mgr.get_transitions(obs0, states, actions, rewards, dones, expl)
    bufs = []
    for _ in range(2):
        a, b = GraphReplayBuffer(64, 5, prioritized=True, alpha=0.5), GraphReplayBuffer(64, 5, prioritized=True, alpha=0.5)
        a.put(maker[:40]); b.put(breaker[:40])
        a.update_priorities(torch.arange(8, device="cuda"), torch.linspace(0.5, 9.0, 8, device="cuda"))
        bufs.append((a, b))
    (a1, b1), (a2, b2) = bufs
    g1 = torch.Generator(device="cuda").manual_seed(5)
    g2 = torch.Generator(device="cuda").manual_seed(5)
    want = a1.sample(16, beta=0.6, generator=g1)
    pend = a2.sample_begin(16, beta=0.6, generator=g2)
    other = b2.sample_begin(16, beta=0.6)                     # interleaved work on the other buffer
    b2.update_priorities(other[0], torch.ones(16, device="cuda"))
    b2.sample_end(other)
    got = a2.sample_end(pend)
    for w_, g_ in zip(want, got):
        if torch.is_tensor(w_):
            assert torch.equal(w_, g_)
        else:
            assert torch.equal(w_.x, g_.x) and torch.equal(w_.edge_index, g_.edge_index) and torch.equal(w_.ptr, g_.ptr)
            assert w_.x._hex_is_maker == g_.x._hex_is_maker and w_.x._hex_max_nodes == g_.x._hex_max_nodes
    pend = a2.sample_begin(8, beta=0.6)
    a2.put(maker[40:44])
    with pytest.raises(RuntimeError, match="between sample_begin and sample_end"):
        a2.sample_end(pend)


def test_sampled_batch_drives_the_model():
    from helpers import make_pair
    from gnn_hex_amd.multi_env_manager import Env_manager
    from gnn_hex_amd.replay import GraphReplayBuffer
    rng = np.random.default_rng(5)
    mgr = Env_manager(8, 7, gamma=0.97, n_steps=[1])
    obs0, states, actions, rewards, dones, expl = _play(mgr, 8, rng)
    maker, _ = mgr.get_transitions(obs0, states, actions, rewards, dones, expl)
    buf = GraphReplayBuffer(128, 7, prioritized=False)
    buf.put(maker)
    hip, ref = make_pair(4, 35, seed=4)
    idx, w, s, s2, a, r, d = buf.sample(8)
    q = hip.simple_forward(s)
    q_ref = ref(s.x.cpu(), s.edge_index.cpu(), s.batch.cpu(), s.ptr.cpu())
    assert (q.detach().cpu() - q_ref.detach()).abs().max() < 1e-4
    qa = q[s.ptr[:-1] + a]                             # Q(s, a): action = node rank inside its graph
    loss = ((qa - r) ** 2 * w).mean()
    loss.backward()
    assert all(p.grad is not None for k, p in hip.named_parameters() if k.startswith("gnn"))


@pytest.mark.parametrize("prune", [True, False])
def test_vectorised_transition_assembly_equals_list_form(prune):
    """assemble_transitions + put_block must leave the replay ring exactly as get_transitions + put does, including
    exploratory pruning, terminal transitions pointing at the start position and multi-n-step ordering."""
    from gnn_hex_amd.multi_env_manager import Env_manager
    from gnn_hex_amd.replay import GraphReplayBuffer
    rng = np.random.default_rng(9)
    mgr = Env_manager(10, 5, gamma=0.9, n_steps=[1, 2], prune_exploratories=prune)
    obs0, states, actions, rewards, dones, expl = _play(mgr, 16, rng)
    expl = [rng.random(10) < 0.2 for _ in range(16)]
    mb, bb = mgr.assemble_transitions(obs0, states, actions, rewards, dones, expl)
    maker, breaker = mgr.get_transitions(obs0, states, actions, rewards, dones, expl)
    assert len(mb) == len(maker) and len(bb) == len(breaker) and len(mb) > 20
    for block, lst in ((mb, maker), (bb, breaker)):
        assert block.action.tolist() == [int(t[1]) for t in lst]
        assert np.allclose(block.reward, [t[2] for t in lst])
        assert block.done.tolist() == [bool(t[4]) for t in lst]
        a, b = GraphReplayBuffer(512, 5, prioritized=False), GraphReplayBuffer(512, 5, prioritized=False)
        a.put_block(block)
        b.put(lst)
        k = len(lst)
        assert len(a) == len(b) == k
        for name in ("adj", "alive", "side"):
            va, vb = getattr(a, name), getattr(b, name)
            assert torch.equal(va[:k], vb[:k]) and torch.equal(va[512:512 + k], vb[512:512 + k]), name
        for name in ("action", "reward", "done"):
            assert torch.equal(getattr(a, name)[:k], getattr(b, name)[:k]), name
        assert np.array_equal(a.n_nodes, b.n_nodes) and np.array_equal(a.n_edges, b.n_edges)
