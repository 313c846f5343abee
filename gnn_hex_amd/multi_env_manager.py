"""Drop-in mirror of ``graph_game.multi_env_manager.Env_manager`` (graph_game/multi_env_manager.py:16-165) on the
HIP board-graph builder (gnn_hex_amd/csrc/env.hip, C ABI ``hexgnn_env_*``).

Same constructor, attributes and methods; ``step`` returns ``(states, rewards, dones, infos)`` with the reference's
reward / reset / side-to-move conventions.  ``states`` is an ``ObsList``: it behaves as the reference's
``List[Data]`` (len / index / iterate -> ``Data(x, edge_index, backmap)`` views) but is backed by ONE batched device
observation, so ``Batch.from_data_list(states)`` costs nothing and carries the sorted CSR the model kernels consume.
All game logic runs on the GPU; there is no python/CPU game fallback.  ``cnn_rep=True`` (CNN input planes) is outside
the hot path and raises NotImplementedError.
"""
from __future__ import annotations

import ctypes as C
import time
from typing import List, Tuple, Union

import numpy as np
import torch

from . import _lib, ops
from .data import Batch, Data


class ObsList:
    """List[Data]-compatible view of one batched observation of all envs."""

    def __init__(self, x, edge_local, edge_global, backmap, batch_vec, node_off, edge_off, gs, is_maker, max_nodes,
                 snap=None):
        self._snap = snap          # (adj [k,nv,W] i64, alive [k,nv] u8) board snapshot the replay buffer stores
        self.x, self.edge_local, self.edge_global, self.backmap = x, edge_local, edge_global, backmap
        self.batch_vec = batch_vec
        self.node_off, self.edge_off = node_off, edge_off     # host lists, len num_envs+1
        self.gs, self.is_maker, self.max_nodes = gs, is_maker, max_nodes
        self._ptr = None
        self._items = {}

    def __len__(self):
        return len(self.node_off) - 1

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        d = self._items.get(i)
        if d is None:
            n0, n1 = self.node_off[i], self.node_off[i + 1]
            e0, e1 = self.edge_off[i], self.edge_off[i + 1]
            d = Data(x=self.x[n0:n1], edge_index=self.edge_local[:, e0:e1], backmap=self.backmap[n0:n1])
            d.x._hex_is_maker = self.is_maker
            d.x._hex_max_nodes = n1 - n0
            d.x._hex_hint_version = d.x._version
            d._hex_src = (self, i)
            self._items[i] = d
        return d

    def snapshot(self):
        if self._snap is None:
            raise ValueError("this observation was taken without board snapshots (Env_manager.record_snapshots=False)")
        return self._snap

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def to_batch(self) -> Batch:
        """The whole observation as a ``Batch`` (what ``Batch.from_data_list(list(states))`` would build)."""
        b = Batch()
        if self._ptr is None:
            self._ptr = torch.tensor(self.node_off, dtype=torch.long, device=self.x.device)
        b.x, b.edge_index, b.batch, b.ptr, b.backmap = self.x, self.edge_global, self.batch_vec, self._ptr, self.backmap
        b._num_graphs = len(self)
        b.x._hex_is_maker = self.is_maker
        b.x._hex_max_nodes = self.max_nodes
        b.x._hex_hint_version = b.x._version
        b.edge_index._hex_csr = self.gs
        return b


class TransitionBlock:
    """Array form of a list of transitions ``(s, a, r, s', done)`` of ONE side: states are referenced as
    ``(index into obs_list, env)`` instead of being materialised as ``Data`` objects; ``next_step == -1`` means the
    start position (terminal transitions, multi_env_manager.py:150-158).  Consumed by ``GraphReplayBuffer.put_block``."""

    def __init__(self, obs_list, start_obs, maker_side, src_step, env, action, reward, next_step, done):
        self.obs_list, self.start_obs, self.maker_side = obs_list, start_obs, maker_side
        self.src_step, self.env, self.action = src_step, env, action
        self.reward, self.next_step, self.done = reward, next_step, done

    def __len__(self):
        return int(self.env.shape[0])


class SnapObs(ObsList):
    """An observation known only as board snapshots + graph sizes (what a device rollout records per step).  It carries
    everything the replay path needs (``snapshot()``, ``node_off`` / ``edge_off``, ``is_maker``); the ``Data`` views of
    an ``ObsList`` are not materialised."""

    def __init__(self, snap, sizes: np.ndarray, is_maker: bool, owner=None):
        self._owner = owner        # (DeviceRollout, run number): the snapshot is a view into that rollout's ring
        node_off = np.zeros(sizes.shape[0] + 1, dtype=np.int64)
        edge_off = np.zeros(sizes.shape[0] + 1, dtype=np.int64)
        np.cumsum(sizes[:, 0], out=node_off[1:])
        np.cumsum(sizes[:, 1], out=edge_off[1:])
        super().__init__(None, None, None, None, None, node_off.tolist(), edge_off.tolist(), None, is_maker,
                         int(sizes[:, 0].max()) if sizes.shape[0] else 0, snap)

    def snapshot(self):
        if self._owner is not None and self._owner[0]._run_no != self._owner[1]:
            raise RuntimeError("stale rollout observation: DeviceRollout.run() has been called again and overwrote the "
                               "snapshot ring this observation views; store a run's transitions (put_block) before the "
                               "next run, or keep RolloutResult.detach() copies")
        return super().snapshot()

    def __getitem__(self, i):
        raise TypeError("a rollout observation holds board snapshots only; re-observe it through GraphReplayBuffer")

    def to_batch(self):
        raise TypeError("a rollout observation holds board snapshots only; re-observe it through GraphReplayBuffer")


class RolloutResult:
    """Histories of one ``DeviceRollout.run()``: ``states[0]`` is the observation the first move was chosen from,
    ``states[t+1]`` the one after move ``t``; ``actions`` are node ranks inside ``states[t]`` (the model's output index,
    what the reference stores), ``vertices`` the vertex ids played.  Host arrays; the snapshots stay on the device."""

    def __init__(self, states, actions, vertices, rewards, dones, exploratories, infos):
        self.states, self.actions, self.vertices = states, actions, vertices
        self.rewards, self.dones, self.exploratories, self.infos = rewards, dones, exploratories, infos

    def detach(self) -> "RolloutResult":
        """Give the states their own copies of the board snapshots.  By default they VIEW the rollout's snapshot ring,
        which the next ``run()`` overwrites (using such a state afterwards raises); a detached result stays valid."""
        for st in self.states:
            if st._owner is not None:
                st._snap = tuple(t.clone() for t in st._snap)
                st._owner = None
        return self


class DeviceRollout:
    """``steps`` lock-step moves of every env of an ``Env_manager`` as a closed device loop: observation (HIP builder) ->
    Q-network forward (advantages only) -> epsilon-greedy -> env step with dead/captured removal and auto reset -> graph
    sizes prefix-summed on the device -> board snapshot.  Nothing is read back between the moves; with ``graph=True`` the
    whole sequence is ONE HIP graph (``gnn_hex_amd.graphs``), so a rollout costs one launch + one read-back.  The
    reference does this with a Python loop over envs per move (graph_game/multi_env_manager.py:76-103 +
    GN0/RainbowDQN/evaluate_elo.py:253-275).  ``steps`` must be even so the side to move is the same before and after;
    the model's weights are read at run time (training between runs is fine, ``grow_*`` needs a new rollout)."""

    def __init__(self, mgr: "Env_manager", model, steps: int = 16, eps: float = 0.05, graph: bool = True):
        if steps < 2 or steps % 2:
            raise ValueError("steps must be a positive even number")
        norms = getattr(getattr(model, "gnn", None), "norms", None)
        # --norm=True normalises over the LIVE batch (all nodes of the current observation).  The rollout hands the model
        # capacity-sized buffers (num_envs * nv rows) whose tail rows are stale once nodes have been removed: the norm kernels
        # take the live node total from the device-side prefix sums instead (ops.live_rows).
        self._live_norm = norms is not None
        if norms is not None and any(type(m).__name__ != "LayerNorm" for m in norms):
            raise NotImplementedError("DeviceRollout supports the whole-batch LayerNorm of --norm=True; per-channel "
                                      "CachedGraphNorm statistics need exact-size batches: use Env_manager.observe()/step()")
        self.mgr, self.model, self.T, self.eps = mgr, model, steps, float(eps)
        self.start_side = mgr.global_onturn
        dev = mgr.device
        k, nv, W = mgr.num_envs, mgr._nv, mgr._words
        e_max = int(mgr._base_sizes[0, 1])
        N, E = k * nv, k * e_max
        self.x = torch.zeros((N, 3), dtype=torch.float32, device=dev)
        self.backmap = torch.zeros(N, dtype=torch.long, device=dev)
        self.batch_vec = torch.zeros(N, dtype=torch.long, device=dev)
        self.edge_local = torch.zeros((2, E), dtype=torch.long, device=dev)
        self.edge_global = torch.zeros((2, E), dtype=torch.long, device=dev)
        rowptr = torch.zeros(N + 1, dtype=torch.int32, device=dev)
        col = torch.zeros(E, dtype=torch.int32, device=dev)
        invdeg = torch.ones(N, dtype=torch.float32, device=dev)
        self.gs = ops.GraphStructure.from_csr(N, E, rowptr, col, invdeg)
        self.node_off = torch.zeros(k + 1, dtype=torch.int32, device=dev)
        self.edge_off = torch.zeros(k + 1, dtype=torch.int32, device=dev)
        T = steps
        self.vert = torch.zeros((T, k), dtype=torch.int32, device=dev)
        self.rank = torch.zeros((T, k), dtype=torch.int32, device=dev)
        self.expl = torch.zeros((T, k), dtype=torch.uint8, device=dev)
        self.result = torch.zeros((T, k, 5), dtype=torch.int32, device=dev)
        self.uni = torch.zeros((T, k, 2), dtype=torch.float32, device=dev)
        self.adj = torch.zeros((T + 1, k, nv, W), dtype=torch.int64, device=dev)
        self.alive = torch.zeros((T + 1, k, nv), dtype=torch.uint8, device=dev)
        self._mt = torch.zeros(k, dtype=torch.int32, device=dev)
        self._tm = torch.zeros(k, dtype=torch.int32, device=dev)
        self._iota = torch.arange(N + 1, dtype=torch.int32, device=dev) if nv > 128 or self._live_norm else None
        self._run_no = 0           # bumped by every run(): results of earlier runs view overwritten snapshots
        self._graph = None
        if graph:
            from .graphs import GraphedStep
            self._upload_offsets()
            state = self.mgr._state_tensors()
            self._graph = GraphedStep(self._body, warmup=1)
            self.mgr._restore_state_tensors(state)      # warm-up + capture played moves: put the boards back

    # -- pieces ----------------------------------------------------------------------------------------
    def _upload_offsets(self):
        sizes = self.mgr._sizes
        k = self.mgr.num_envs
        off = np.zeros(2 * (k + 1), dtype=np.int32)
        np.cumsum(sizes[:, 0], out=off[1:k + 1])
        np.cumsum(sizes[:, 1], out=off[k + 2:])
        t = torch.from_numpy(off).to(self.mgr.device)
        self.node_off.copy_(t[:k + 1])
        self.edge_off.copy_(t[k + 1:])

    def _export(self, slot: int):
        _lib.check(_lib.lib().hexgnn_env_export(self.mgr._h, self.adj[slot].data_ptr(), self.alive[slot].data_ptr(),
                                               self._mt.data_ptr(), self._tm.data_ptr(), None, None, ops._stream()),
                   "hexgnn_env_export")

    def _body(self):
        L = _lib.lib()
        mgr, k = self.mgr, self.mgr.num_envs
        E = int(self.edge_global.shape[1])
        self._export(0)
        maker = self.start_side == "m"
        for t in range(self.T):
            _lib.check(L.hexgnn_env_observe(mgr._h, self.node_off.data_ptr(), self.edge_off.data_ptr(), E,
                                            self.x.data_ptr(), self.backmap.data_ptr(), self.edge_local.data_ptr(),
                                            self.edge_global.data_ptr(), self.gs.rowptr.data_ptr(), self.gs.col.data_ptr(),
                                            self.gs.invdeg.data_ptr(), self.batch_vec.data_ptr(), ops._stream()),
                       "hexgnn_env_observe")
            if self._iota is not None:
                # Boards above 128 nodes (and --norm=True models) run on the layer-major kernels, which walk ALL rows of the capacity-sized buffers:
                # rows past the current total must be empty, not whatever an earlier, larger observation left there (the
                # row right behind the end marker otherwise shows a bogus degree of thousands: 146 us per layer launch).
                torch.where(self._iota > self.node_off[k], self.edge_off[k], self.gs.rowptr, out=self.gs.rowptr)
            x = self.x.view(self.x.shape)       # fresh tensor object per step: the hints below differ per side
            x._hex_is_maker = maker
            x._hex_max_nodes = mgr._nv
            x._hex_hint_version = x._version
            ei = self.edge_global.view(self.edge_global.shape)
            ei._hex_csr = self.gs
            with torch.no_grad(), ops.live_rows(self.node_off[k:k + 1] if self._live_norm else None):
                adv = self.model(x, ei, self.batch_vec, self.node_off, advantages_only=True)
            u = None
            if self.eps > 0:
                u = self.uni[t]
                u.uniform_()
            _lib.check(L.hexgnn_select_actions(k, self.node_off.data_ptr(), adv.reshape(-1).data_ptr(),
                                               self.backmap.data_ptr(), self.eps, u.data_ptr() if u is not None else None,
                                               self.vert[t].data_ptr(), self.rank[t].data_ptr(), self.expl[t].data_ptr(),
                                               ops._stream()), "hexgnn_select_actions")
            # finished envs restart on the side everyone moves to next (multi_env_manager.py:98-101)
            _lib.check(L.hexgnn_env_step(mgr._h, self.vert[t].data_ptr(), 1, 1, int(not maker), self.result[t].data_ptr(),
                                         ops._stream()), "hexgnn_env_step")
            _lib.check(L.hexgnn_env_offsets(k, self.result[t].data_ptr(), self.node_off.data_ptr(),
                                            self.edge_off.data_ptr(), ops._stream()), "hexgnn_env_offsets")
            self._export(t + 1)
            maker = not maker

    # -- public ----------------------------------------------------------------------------------------
    def run(self) -> RolloutResult:
        return self.run_end(self.run_begin())

    def run_begin(self):
        """Issue the rollout (stream-ordered, nothing is waited for) and return a handle for ``run_end``.  Work issued
        between the two calls runs behind the rollout on the GPU while the host is free -- e.g. a learner that starts the
        next rollout before it issues this iteration's updates (the actor then plays with weights one iteration old).
        The previous run's observations must have been stored (``put_block``) by now: this run overwrites their ring."""
        mgr = self.mgr
        if mgr.global_onturn != self.start_side:
            raise RuntimeError("this rollout was built with %r to move" % self.start_side)
        if getattr(self, "_open", False):
            raise RuntimeError("run_begin() called twice without run_end()")
        sizes0 = mgr._sizes.copy()
        self._run_no += 1
        self._upload_offsets()
        if self._graph is not None:
            self._graph.replay()
        else:
            self._body()
        self._open = True
        return (self._run_no, sizes0)

    def run_end(self, handle) -> RolloutResult:
        mgr = self.mgr
        run_no, sizes0 = handle
        if not getattr(self, "_open", False) or run_no != self._run_no:
            raise RuntimeError("run_end() needs the handle of the rollout issued last")
        self._open = False
        owner = (self, run_no)
        res = self.result.cpu().numpy()                      # the only read-back (also the synchronisation point)
        if res[:, :, 4].any():
            t, i = np.argwhere(res[:, :, 4])[0]
            raise ValueError("illegal action at step %d for env %d" % (t, i))
        ranks = self.rank.cpu().numpy().astype(np.int64)
        verts = self.vert.cpu().numpy().astype(np.int64)
        expl = self.expl.cpu().numpy().astype(bool)
        T, k = self.T, mgr.num_envs
        dones = res[:, :, 0] >= 0
        rewards = np.zeros((T, k), dtype=float)
        infos = [[{} for _ in range(k)] for _ in range(T)]
        now = time.perf_counter()
        maker = self.start_side == "m"
        states = [SnapObs((self.adj[0], self.alive[0]), sizes0, maker, owner)]
        for t in range(T):
            mover_is_maker = maker
            for i in np.nonzero(dones[t])[0]:
                winner_is_maker = res[t, i, 0] == 0
                n = int(res[t, i, 1])
                infos[t][i]["episode_metrics"] = {
                    "return": 1 if winner_is_maker else -1,
                    "discounted_return": float(mgr.gamma ** n if winner_is_maker else -(mgr.gamma ** n)),
                    "length": n,
                    "time": now - mgr._creation_time[i],
                }
                rewards[t, i] = 1 if winner_is_maker == mover_is_maker else -1
                mgr._creation_time[i] = now
            maker = not maker
            states.append(SnapObs((self.adj[t + 1], self.alive[t + 1]), res[t, :, 2:4].astype(np.int64), maker, owner))
        mgr._sizes = res[-1, :, 2:4].astype(np.int64)
        mgr.last_obs = None
        return RolloutResult(states, ranks, verts, rewards, dones, expl, infos)


class RolloutStitcher:
    """Transitions across rollout boundaries.  An n-step transition that starts in move i needs the histories up to move
    i + 2 n_step, so the last 2 n_step - 1 moves of a rollout cannot produce transitions on their own (3 of 16 moves at
    n_step = 2).  The stitcher keeps that tail -- observations as detached snapshot copies, actions, rewards, dones,
    exploratory flags -- and prepends it to the next rollout's histories: every move of a continuous self-play stream then
    starts exactly one transition per n_step, and none twice (windows of a shorter n_step that were already complete in the
    previous rollout are skipped).

        stitch = RolloutStitcher(mgr)
        maker_block, breaker_block = stitch.assemble(rollout.run())      # instead of mgr.assemble_transitions(...)
    """

    def __init__(self, mgr: "Env_manager"):
        self.mgr = mgr
        self.keep = 2 * max(mgr.n_steps) - 1     # moves whose longest window was still incomplete at the rollout's end
        self._tail = None      # (states [keep + 1], actions, rewards, dones, exploratories) of the previous rollout's end

    def reset(self):
        """Forget the tail (after ``mgr.reset()`` / ``change_hex_size``: the stream is no longer continuous)."""
        self._tail = None

    def assemble(self, res: RolloutResult):
        T = len(res.actions)
        if T < self.keep:
            raise ValueError("a rollout must be at least 2 * n_step - 1 = %d moves long" % self.keep)
        states = list(res.states)
        acts, rews = list(res.actions), list(res.rewards)
        dones, expl = list(res.dones), list(res.exploratories)
        emitted = None
        if self._tail is not None:
            t_states, t_acts, t_rews, t_dones, t_expl = self._tail
            # the tail's last observation IS this rollout's first one (same boards): keep the tail's copy for the prefix
            states = t_states[:-1] + states
            acts, rews, dones, expl = t_acts + acts, t_rews + rews, t_dones + dones, t_expl + expl
            # a shorter window (n_step below the maximum) starting in the first 2 (n_max - n_step) prefix moves was already
            # complete in the previous rollout
            emitted = {n: self.keep + 1 - 2 * n for n in self.mgr.n_steps}
        blocks = self.mgr.assemble_transitions(states[0], states[1:], acts, rews, dones, expl, _already_emitted=emitted)
        # next prefix: the last `keep` moves with their keep + 1 observations, snapshots copied out of the rollout's ring
        k = self.keep
        tail_states = []
        for st in states[-(k + 1):]:
            if isinstance(st, SnapObs) and st._owner is not None:
                cp = SnapObs.__new__(SnapObs)
                cp.__dict__.update(st.__dict__)
                cp._snap = tuple(t.clone() for t in st._snap)
                cp._owner = None
                cp._items = {}
                st = cp
            tail_states.append(st)
        self._tail = (tail_states, acts[-k:], rews[-k:], dones[-k:], expl[-k:])
        return blocks


class _EnvView:
    """Read-only stand-in for the per-env ``Hex_game`` objects of the reference's ``Env_manager.envs``."""

    def __init__(self, mgr, idx):
        self._m, self._i = mgr, idx

    @property
    def onturn(self):
        return self._m.global_onturn

    @property
    def not_onturn(self):
        return "b" if self._m.global_onturn == "m" else "m"

    @property
    def total_num_moves(self):
        return int(self._m._state()["total_moves"][self._i])

    @property
    def creation_time(self):
        return self._m._creation_time[self._i]

    def get_actions(self):
        return self._m.get_valid_actions()[self._i]


class Env_manager:
    last_obs: ObsList

    def __init__(self, num_envs, hex_size, gamma=1, n_steps=[1], prune_exploratories=True, cnn_rep=False,
                 cnn_hex_size=5, gao_mode=False, border_fill=True, device=None):
        if cnn_rep:
            raise NotImplementedError("cnn_rep=True (CNN planes) is outside the accelerated hot path")
        self.num_envs = num_envs
        self.gamma = gamma
        self.global_onturn = "m"
        self.n_steps = n_steps
        self.prune_exploratories = prune_exploratories
        self.cnn_rep = cnn_rep
        self.gao_mode = gao_mode
        self.border_fill = border_fill
        self.cnn_hex_size = cnn_hex_size
        self.record_snapshots = True     # keep the 2 KB/env board snapshot with every observation (replay storage)
        self.device = torch.device(device if device is not None else "cuda")
        if self.device.type != "cuda":
            raise _lib.HexGnnError("Env_manager runs only on the MI355X HIP path (no CPU fallback)")
        self._h = None
        self._base = None
        self.change_hex_size(hex_size)

    # ---- handles ---------------------------------------------------------------------------------------
    def _destroy(self):
        L = _lib.lib()
        for name in ("_h", "_base"):
            h = getattr(self, name, None)
            if h:
                L.hexgnn_env_destroy(h)
                setattr(self, name, None)

    def __del__(self):
        try:
            self._destroy()
        except Exception:
            pass

    def change_hex_size(self, new_size):
        L = _lib.lib()
        self._destroy()
        self.hex_size = new_size
        self.global_onturn = "m"
        with torch.cuda.device(self.device):
            h, hb = C.c_void_p(), C.c_void_p()
            _lib.check(L.hexgnn_env_create(self.num_envs, new_size, C.byref(h)), "hexgnn_env_create")
            _lib.check(L.hexgnn_env_create(1, new_size, C.byref(hb)), "hexgnn_env_create")
        self._h, self._base = h, hb
        self._nv = L.hexgnn_env_num_vertices(h)
        self._words = L.hexgnn_env_words(h)
        nv, n = self._nv, new_size
        e_start = 2 * (2 * n + (n - 2) * (n - 1) + n * (n - 1) + (n - 1) ** 2)
        self._sizes = np.tile(np.array([[nv, e_start]], dtype=np.int64), (self.num_envs, 1))
        self._base_sizes = np.array([[nv, e_start]], dtype=np.int64)
        self._creation_time = [time.perf_counter()] * self.num_envs
        self.envs = [_EnvView(self, i) for i in range(self.num_envs)]
        self.base_game = None
        self.last_obs = None

    # ---- observation -----------------------------------------------------------------------------------
    def _snapshot_handle(self, h, k):
        L = _lib.lib()
        dev = self.device
        adj = torch.empty((k, self._nv, self._words), dtype=torch.int64, device=dev)
        alive = torch.empty((k, self._nv), dtype=torch.uint8, device=dev)
        mt = torch.empty(k, dtype=torch.int32, device=dev)
        tm = torch.empty(k, dtype=torch.int32, device=dev)
        _lib.check(L.hexgnn_env_export(h, adj.data_ptr(), alive.data_ptr(), mt.data_ptr(), tm.data_ptr(), None, None,
                                       ops._stream()), "hexgnn_env_export")
        return adj, alive

    def _observe_handle(self, h, sizes, is_maker) -> ObsList:
        L = _lib.lib()
        dev = self.device
        k = sizes.shape[0]
        node_off = np.zeros(k + 1, dtype=np.int64)
        edge_off = np.zeros(k + 1, dtype=np.int64)
        np.cumsum(sizes[:, 0], out=node_off[1:])
        np.cumsum(sizes[:, 1], out=edge_off[1:])
        N, E = int(node_off[-1]), int(edge_off[-1])
        offs = torch.from_numpy(np.concatenate([node_off, edge_off]).astype(np.int32)).to(dev, non_blocking=True)
        x = torch.empty((N, 3), dtype=torch.float32, device=dev)
        backmap = torch.empty(N, dtype=torch.long, device=dev)
        batch_vec = torch.empty(N, dtype=torch.long, device=dev)
        edge_local = torch.empty((2, max(E, 1)), dtype=torch.long, device=dev)[:, :E]
        edge_global = torch.empty((2, max(E, 1)), dtype=torch.long, device=dev)[:, :E]
        rowptr = torch.empty(N + 1, dtype=torch.int32, device=dev)
        col = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
        invdeg = torch.empty(max(N, 1), dtype=torch.float32, device=dev)
        if E == 0:
            edge_local = torch.empty((2, 0), dtype=torch.long, device=dev)
            edge_global = torch.empty((2, 0), dtype=torch.long, device=dev)
        el_base = edge_local.data_ptr() if E > 0 else x.data_ptr()
        eg_base = edge_global.data_ptr() if E > 0 else x.data_ptr()
        _lib.check(L.hexgnn_env_observe(h, offs.data_ptr(), offs[k + 1:].data_ptr(), E, x.data_ptr(), backmap.data_ptr(),
                                        el_base, eg_base, rowptr.data_ptr(), col.data_ptr(), invdeg.data_ptr(),
                                        batch_vec.data_ptr(), ops._stream()), "hexgnn_env_observe")
        gs = ops.GraphStructure.from_csr(N, E, rowptr, col, invdeg)
        snap = self._snapshot_handle(h, k) if self.record_snapshots else None
        return ObsList(x, edge_local, edge_global, backmap, batch_vec, node_off.tolist(), edge_off.tolist(), gs,
                       is_maker, int(sizes[:, 0].max()) if k else 0, snap)

    @property
    def starting_obs(self) -> Data:
        """Fresh ``Data`` of the start position with maker to move (multi_env_manager.py:41-49)."""
        obs = self._observe_handle(self._base, self._base_sizes, True)
        d = obs[0]
        out = Data(x=d.x.clone(), edge_index=d.edge_index.clone(), backmap=d.backmap.clone())
        out._hex_src = (obs, 0)
        return out

    def observe(self) -> ObsList:
        self.last_obs = self._observe_handle(self._h, self._sizes, self.global_onturn == "m")
        return self.last_obs

    # ---- actions ---------------------------------------------------------------------------------------
    @staticmethod
    def validate_actions(states, actions: List[int]):
        """Node rank -> vertex id through each state's backmap (multi_env_manager.py:62-64)."""
        if isinstance(states, ObsList):
            idx = torch.as_tensor([int(a) for a in actions], dtype=torch.long, device=states.x.device)
            base = torch.as_tensor(states.node_off[:-1], dtype=torch.long, device=states.x.device)
            return states.backmap[base + idx].tolist()
        return [state.backmap[action].item() for state, action in zip(states, actions)]

    def select_actions(self, q: torch.Tensor, states: "ObsList" = None, eps: float = 0.0, generator=None):
        """Device-side acting: epsilon-greedy over each env's non-terminal nodes given the model output for the batched
        observation (``Q`` or ``advantages_only`` values).  Returns ``(vertex_actions int32 [num_envs] on the device --
        ready for ``step`` --, node ranks, exploratory flags)``; nothing is copied to the host."""
        states = states if states is not None else self.last_obs
        dev = self.device
        k = len(states)
        if states._ptr is None:
            states._ptr = torch.tensor(states.node_off, dtype=torch.long, device=dev)
        gptr = states._ptr.to(torch.int32)
        qf = q.reshape(-1).float().contiguous()
        vert = torch.empty(k, dtype=torch.int32, device=dev)
        rank = torch.empty(k, dtype=torch.int32, device=dev)
        expl = torch.empty(k, dtype=torch.uint8, device=dev)
        u = torch.rand((k, 2), dtype=torch.float32, device=dev, generator=generator) if eps > 0 else None
        _lib.check(_lib.lib().hexgnn_select_actions(k, gptr.data_ptr(), qf.data_ptr(), states.backmap.data_ptr(),
                                                    float(eps), u.data_ptr() if u is not None else None, vert.data_ptr(),
                                                    rank.data_ptr(), expl.data_ptr(), ops._stream()),
                   "hexgnn_select_actions")
        return vert, rank, expl.bool()

    def get_valid_actions(self) -> List[np.ndarray]:
        obs = self.last_obs if self.last_obs is not None else self.observe()
        bm = obs.backmap.cpu().numpy()
        return [bm[obs.node_off[i] + 2:obs.node_off[i + 1]] for i in range(self.num_envs)]

    def sample(self) -> np.ndarray:
        return np.array([np.random.choice(x) for x in self.get_valid_actions()])

    # ---- stepping --------------------------------------------------------------------------------------
    def step(self, actions) -> Tuple[ObsList, np.ndarray, np.ndarray, List[dict]]:
        L = _lib.lib()
        if torch.is_tensor(actions):
            act = actions.to(device=self.device, dtype=torch.int32).contiguous()
        else:
            act = torch.as_tensor(np.asarray([int(a) for a in actions], dtype=np.int32)).to(self.device)
        if act.numel() != self.num_envs:
            raise ValueError("expected %d actions" % self.num_envs)
        result = torch.empty((self.num_envs, 5), dtype=torch.int32, device=self.device)
        reset_maker = self.global_onturn == "b"      # finished envs restart on the side everyone moves to next
        _lib.check(L.hexgnn_env_step(self._h, act.data_ptr(), 1, 1, int(reset_maker), result.data_ptr(), ops._stream()),
                   "hexgnn_env_step")
        res = result.cpu().numpy()
        if res[:, 4].any():
            bad = int(np.nonzero(res[:, 4])[0][0])
            raise ValueError("illegal action %d for env %d" % (int(act[bad]), bad))
        now = time.perf_counter()
        rewards = np.zeros(self.num_envs, dtype=float)
        dones = res[:, 0] >= 0
        infos = [{} for _ in range(self.num_envs)]
        mover = self.global_onturn
        for i in np.nonzero(dones)[0]:
            winner = "m" if res[i, 0] == 0 else "b"
            n = int(res[i, 1])
            infos[i]["episode_metrics"] = {
                "return": 1 if winner == "m" else -1,
                "discounted_return": float(1 * self.gamma ** n if winner == "m" else -1 * self.gamma ** n),
                "length": n,
                "time": now - self._creation_time[i],
            }
            rewards[i] = 1 if winner == mover else -1       # winner == env.not_onturn after the move
            self._creation_time[i] = now
        self.global_onturn = "m" if self.global_onturn == "b" else "b"
        self._sizes = res[:, 2:4].astype(np.int64)
        states = self.observe()
        return states, rewards, dones.astype(bool), infos

    def reset(self) -> ObsList:
        L = _lib.lib()
        self.global_onturn = "m"
        _lib.check(L.hexgnn_env_reset(self._h, None, 1, None, ops._stream()), "hexgnn_env_reset")
        self._sizes = np.tile(self._base_sizes, (self.num_envs, 1))
        self._creation_time = [time.perf_counter()] * self.num_envs
        return self.observe()

    def _state_tensors(self):
        """Device copy of the full env state (boards, side, move counters, response sets)."""
        L = _lib.lib()
        dev = self.device
        k, nv = self.num_envs, self._nv
        t = dict(adj=torch.empty((k, nv, self._words), dtype=torch.int64, device=dev),
                 alive=torch.empty((k, nv), dtype=torch.uint8, device=dev),
                 mt=torch.empty(k, dtype=torch.int32, device=dev), tm=torch.empty(k, dtype=torch.int32, device=dev),
                 rm=torch.empty((k, nv), dtype=torch.int16, device=dev), rb=torch.empty((k, nv), dtype=torch.int16, device=dev))
        _lib.check(L.hexgnn_env_export(self._h, t["adj"].data_ptr(), t["alive"].data_ptr(), t["mt"].data_ptr(),
                                       t["tm"].data_ptr(), t["rm"].data_ptr(), t["rb"].data_ptr(), ops._stream()),
                   "hexgnn_env_export")
        t["sizes"], t["onturn"] = self._sizes.copy(), self.global_onturn
        return t

    def _restore_state_tensors(self, t):
        _lib.check(_lib.lib().hexgnn_env_import(self._h, t["adj"].data_ptr(), t["alive"].data_ptr(), t["mt"].data_ptr(),
                                               t["tm"].data_ptr(), t["rm"].data_ptr(), t["rb"].data_ptr(), ops._stream()),
                   "hexgnn_env_import")
        self._sizes, self.global_onturn = t["sizes"].copy(), t["onturn"]
        self.last_obs = None

    # ---- raw state (tests, debugging) ------------------------------------------------------------------
    def _state(self):
        L = _lib.lib()
        dev = self.device
        adj = torch.empty((self.num_envs, self._nv, self._words), dtype=torch.int64, device=dev)
        alive = torch.empty((self.num_envs, self._nv), dtype=torch.uint8, device=dev)
        mt = torch.empty(self.num_envs, dtype=torch.int32, device=dev)
        tm = torch.empty(self.num_envs, dtype=torch.int32, device=dev)
        rm = torch.empty((self.num_envs, self._nv), dtype=torch.int16, device=dev)
        rb = torch.empty((self.num_envs, self._nv), dtype=torch.int16, device=dev)
        _lib.check(L.hexgnn_env_export(self._h, adj.data_ptr(), alive.data_ptr(), mt.data_ptr(), tm.data_ptr(),
                                       rm.data_ptr(), rb.data_ptr(), ops._stream()), "hexgnn_env_export")
        return dict(adj=adj.cpu().numpy().view(np.uint64), alive=alive.cpu().numpy(), maker_turn=mt.cpu().numpy(),
                    total_moves=tm.cpu().numpy(), resp_maker=rm.cpu().numpy(), resp_breaker=rb.cpu().numpy())

    # ---- transition assembly, array form (same semantics as get_transitions below, no per-transition objects) --
    def assemble_transitions(self, starting_states, state_history: list, action_history: list, reward_history: list,
                             done_history: list, exploratories_history: list, _already_emitted=None):
        """``get_transitions`` (multi_env_manager.py:113-165) vectorised over the envs: returns
        ``(maker_block, breaker_block)`` of ``TransitionBlock`` whose entries appear in exactly the order the list
        form emits them.  Histories may hold ``ObsList`` observations (states stay on the device as board snapshots)."""
        sh = list(state_history)
        sh.insert(0, starting_states)
        T = len(action_history)
        A = np.asarray([np.asarray(a, dtype=np.int64) for a in action_history]).reshape(T, -1)
        R = np.asarray(reward_history, dtype=np.float64).reshape(T, -1)
        D = np.asarray(done_history, dtype=bool).reshape(T, -1)
        X = np.asarray(exploratories_history, dtype=bool).reshape(T, -1)
        E = A.shape[1]
        ks = np.arange(E)
        out = {True: [], False: []}
        for i in range(T):
            start_state = sh[i]
            maker_side = start_state.is_maker if isinstance(start_state, ObsList) else bool(start_state[0].x[0, 2] == 1)
            for n_step in self.n_steps:
                if not len(sh) > i + 2 * n_step:
                    continue
                if _already_emitted is not None and i < _already_emitted.get(n_step, 0):
                    continue        # RolloutStitcher: this window was complete, and emitted, in the previous rollout
                w = 2 * n_step
                jj = np.arange(w)
                coef = ((-(jj % 2)) * 2 + 1) * (float(self.gamma) ** (jj // 2))          # sign * gamma^((j-i)//2)
                r = R[i:i + w] * coef[:, None]
                csum = np.cumsum(r, axis=0)                                                # reward through window step j
                d = D[i:i + w]
                first_done = np.where(d.any(0), d.argmax(0), w)                           # w = none
                if self.prune_exploratories:
                    x = X[i:i + w].copy()
                    x[0] = False                                                           # only j > i prunes
                    first_expl = np.where(x.any(0), x.argmax(0), w)
                else:
                    first_expl = np.full(E, w)
                terminal = (first_done < w) & (first_done <= first_expl)
                pruned = (first_expl < w) & (first_expl < first_done)
                full = ~terminal & ~pruned
                emit = terminal | full
                jend = np.where(terminal, first_done, w - 1)
                reward = csum[jend, ks]
                nxt = np.where(terminal, -1, i + w)
                sel = ks[emit]
                out[maker_side].append((np.full(sel.shape, i), sel, A[i, sel], reward[sel], nxt[sel], terminal[sel]))
        blocks = []
        start_obs = None
        for side in (True, False):
            if out[side]:
                cat = [np.concatenate(c) for c in zip(*out[side])]
            else:
                cat = [np.zeros(0, dtype=np.int64)] * 3 + [np.zeros(0)] + [np.zeros(0, dtype=np.int64), np.zeros(0, dtype=bool)]
            if start_obs is None and len(cat[1]) and (cat[4] < 0).any():
                start_obs = self._observe_handle(self._base, self._base_sizes, True)
            blocks.append(TransitionBlock(sh, start_obs, side, *cat))
        if start_obs is not None:
            for b in blocks:
                b.start_obs = start_obs
        return blocks[0], blocks[1]

    # ---- transition assembly, list form (the reference's return type) -----------------------------------------
    def get_transitions(self, starting_states, state_history: list, action_history: list, reward_history: list,
                        done_history: list, exploratories_history: list):
        """``(maker_transitions, breaker_transitions)``, each a list of ``(s, a, r, s', done)`` -- the reference's
        n-step, sign-alternating, gamma-discounted two-player transitions, an exploratory action after the first step
        pruning a transition (graph_game/multi_env_manager.py:113-165).  Built from ``assemble_transitions`` (the
        rewards, windows and emission order are computed once, vectorised over envs); this method only materialises the
        per-transition ``Data`` views the list API promises: ``backmap`` removed from the states, a fresh start
        observation with the side column set for terminal transitions."""
        blocks = self.assemble_transitions(starting_states, state_history, action_history, reward_history,
                                           done_history, exploratories_history)
        out = []
        for blk in blocks:
            lst = []
            for src, env, act, rew, nxt, done in zip(blk.src_step.tolist(), blk.env.tolist(), blk.action.tolist(),
                                                     blk.reward.tolist(), blk.next_step.tolist(), blk.done.tolist()):
                s_k = blk.obs_list[src][env]
                if hasattr(s_k, "backmap"):
                    s_k.__delattr__("backmap")
                assert act < len(s_k.x)
                if done:
                    s_next = self.starting_obs
                    s_next.__delattr__("backmap")
                    s_next.x[:, 2] = 1.0 if blk.maker_side else 0.0
                    s_next.x._hex_is_maker = blk.maker_side
                    s_next.x._hex_hint_version = s_next.x._version
                else:
                    s_next = blk.obs_list[nxt][env]
                    if hasattr(s_next, "backmap"):
                        s_next.__delattr__("backmap")
                lst.append((s_k, action_history[src][env], rew, s_next, bool(done)))
            out.append(lst)
        return out[0], out[1]
