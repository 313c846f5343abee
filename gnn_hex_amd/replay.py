"""Device-resident n-step replay for the RainbowDQN loop: transitions as produced by ``Env_manager.get_transitions``
are stored in HBM as fixed-size board-state snapshots (the env's adjacency bit matrix: 2 KB per Hex-11 state), sampled
uniformly or with proportional prioritisation (sum / min trees on the device, ``hexgnn_per_*``), and the sampled
states are rebuilt into ``Batch`` objects by the same observation kernel the env uses (``hexgnn_states_observe``) --
sorted CSR, side to move and largest graph size attached, no host-side collation.

The reference's buffer lives in the un-vendored submodule ``GN0/RainbowDQN/Rainbow`` (.gitmodules:1-4); only its flags
are known (README.md:5,7: ``--buffer_size=260000 --burnin=20000 --prioritized_er=True --prioritized_er_beta0=0.6
--n_step=2``).  PARITY UNPINNED: this class follows the published algorithm (oracle/replay_ref.py) and is checked
against that oracle; method names follow the upstream buffer (``put`` / ``sample`` / ``update_priorities``).
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch

from . import _lib, ops
from .data import Batch


def _pow2_at_least(n: int) -> int:
    p = 1
    while p < n:
        p *= 2
    return p


class GraphReplayBuffer:
    """Ring of ``capacity`` transitions ``(state, action, reward, next_state, done)`` for one side (the loop keeps one
    buffer for maker and one for breaker transitions, multi_env_manager.py:139)."""

    def __init__(self, capacity: int, hex_size: int, prioritized: bool = True, alpha: float = 0.5, eps: float = 1e-6,
                 burnin: int = 0, device="cuda", pack_blocks: bool = True):
        """``pack_blocks`` (boards above 128 nodes, Hex-12 and larger, which run on the one-launch SAGE stack kernels): a draw is
        listed in ``data.pack_order`` order -- indices, weights, both batches, actions, rewards and done flags alike, the order of
        a random draw carries no meaning -- with the row-block tables attached, so that workgroups hold whole graphs or their own
        pieces of a large one instead of whatever a multiple of 128 rows cuts."""
        self.capacity, self.hex_size = int(capacity), int(hex_size)
        self.pack_blocks = bool(pack_blocks) and hex_size * hex_size + 2 > 128
        self._max_blocks = None
        self.prioritized, self.alpha, self.eps, self.burnin = prioritized, float(alpha), float(eps), int(burnin)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.HexGnnError("GraphReplayBuffer lives in HBM (no CPU fallback)")
        nv = hex_size * hex_size + 2
        self.nv, self.words = nv, (nv + 63) // 64
        dev, C = self.device, self.capacity
        z = lambda *shape, dtype: torch.zeros(shape, dtype=dtype, device=dev)
        # states and next states share one array pair: slot i = state of transition i, slot C + i = its next state
        self.adj = z(2 * C, nv, self.words, dtype=torch.int64)
        self.alive = z(2 * C, nv, dtype=torch.uint8)
        self.side = z(2 * C, dtype=torch.uint8)
        self.action = z(C, dtype=torch.long)
        self.reward = z(C, dtype=torch.float32)
        self.done = z(C, dtype=torch.bool)
        self.n_nodes = np.zeros(2 * C, dtype=np.int64)      # host copies of the graph sizes (known when stored)
        self.n_edges = np.zeros(2 * C, dtype=np.int64)
        self.side_host = np.zeros(2 * C, dtype=np.uint8)    # host mirror of `side` (sampling needs it without a sync)
        self.pos, self.size = 0, 0
        self.cap2 = _pow2_at_least(C)
        self.sum_tree = torch.empty(2 * self.cap2, dtype=torch.float64, device=dev)
        self.min_tree = torch.empty(2 * self.cap2, dtype=torch.float64, device=dev)
        _lib.check(_lib.lib().hexgnn_per_init(self.cap2, self.sum_tree.data_ptr(), self.min_tree.data_ptr(), ops._stream()),
                   "hexgnn_per_init")
        self.max_priority = torch.ones((), dtype=torch.float64, device=dev)

    def __len__(self):
        return self.size

    @property
    def burnedin(self) -> bool:
        return self.size >= self.burnin

    # ---- insertion ---------------------------------------------------------------------------------------
    @staticmethod
    def _source(d):
        src = getattr(d, "_hex_src", None)
        if src is None:
            raise ValueError("GraphReplayBuffer stores states observed by gnn_hex_amd.Env_manager (they carry the "
                             "board snapshot); got a Data object without one")
        return src

    def _store_states(self, datas, slots: np.ndarray):
        groups = {}
        for d, slot in zip(datas, slots):
            obs, i = self._source(d)
            g = groups.setdefault(id(obs), (obs, [], [], []))
            g[1].append(i)
            g[2].append(int(slot))
            hint = getattr(d.x, "_hex_is_maker", None)
            g[3].append(int(obs.is_maker if hint is None else hint))
            self.n_nodes[slot] = obs.node_off[i + 1] - obs.node_off[i]
            self.n_edges[slot] = obs.edge_off[i + 1] - obs.edge_off[i]
        dev = self.device
        for obs, idx, slot, side in groups.values():
            it = torch.as_tensor(idx, dtype=torch.long, device=dev)
            st = torch.as_tensor(slot, dtype=torch.long, device=dev)
            snap_adj, snap_alive = obs.snapshot()
            self.adj[st] = snap_adj[it]
            self.alive[st] = snap_alive[it]
            self.side[st] = torch.as_tensor(side, dtype=torch.uint8, device=dev)
            self.side_host[np.asarray(slot, dtype=np.int64)] = np.asarray(side, dtype=np.uint8)

    def put(self, transitions: List[tuple]) -> None:
        """Append transitions ``(state, action, reward, next_state, done)`` as returned by
        ``Env_manager.get_transitions``; new entries get the running maximum priority."""
        k = len(transitions)
        if k == 0:
            return
        if k > self.capacity:
            transitions = transitions[-self.capacity:]
            k = self.capacity
        C = self.capacity
        slots = (self.pos + np.arange(k)) % C
        self._store_states([t[0] for t in transitions], slots)
        self._store_states([t[3] for t in transitions], slots + C)
        dev = self.device
        st = torch.as_tensor(slots, dtype=torch.long, device=dev)
        self.action[st] = torch.as_tensor([int(t[1]) for t in transitions], dtype=torch.long, device=dev)
        self.reward[st] = torch.as_tensor([float(t[2]) for t in transitions], dtype=torch.float32, device=dev)
        self.done[st] = torch.as_tensor([bool(t[4]) for t in transitions], dtype=torch.bool, device=dev)
        self.pos = int((self.pos + k) % C)
        self.size = min(C, self.size + k)
        if self.prioritized:
            self._tree_update_td(st, None)         # new transitions enter at the running maximum priority

    def _store_indexed(self, obs_list, start_obs, steps: np.ndarray, envs: np.ndarray, slots: np.ndarray, side: bool):
        """Copy board snapshots ``obs_list[steps[i]][envs[i]]`` (``steps[i] == -1``: the start position) into ring
        slots: the distinct observations' snapshots are concatenated once and all entries move in ONE indexed copy per
        array (a loop over the observations cost ~8 small launches each: 4 ms of host time per 16-move rollout block)."""
        dev = self.device
        rows = np.zeros(len(steps), dtype=np.int64)      # row of every entry inside the concatenated snapshots
        adjs, alives, off = [], [], 0
        for st in np.unique(steps):
            m = steps == st
            obs = start_obs if st < 0 else obs_list[int(st)]
            idx = np.zeros(int(m.sum()), dtype=np.int64) if st < 0 else envs[m]
            sl = slots[m]
            noff, eoff = np.asarray(obs.node_off), np.asarray(obs.edge_off)
            self.n_nodes[sl] = noff[idx + 1] - noff[idx]
            self.n_edges[sl] = eoff[idx + 1] - eoff[idx]
            snap_adj, snap_alive = obs.snapshot()
            adjs.append(snap_adj)
            alives.append(snap_alive)
            rows[m] = off + idx
            off += snap_adj.shape[0]
        if not adjs:
            return
        src = torch.from_numpy(rows).to(dev)
        dst = torch.from_numpy(np.ascontiguousarray(slots, dtype=np.int64)).to(dev)
        all_adj = adjs[0] if len(adjs) == 1 else torch.cat(adjs)
        all_alive = alives[0] if len(alives) == 1 else torch.cat(alives)
        self.adj.index_copy_(0, dst, all_adj.index_select(0, src))
        self.alive.index_copy_(0, dst, all_alive.index_select(0, src))
        self.side.index_fill_(0, dst, 1 if side else 0)
        self.side_host[slots] = 1 if side else 0

    def put_block(self, block) -> None:
        """Append a ``TransitionBlock`` from ``Env_manager.assemble_transitions`` (array form of ``put``)."""
        k = len(block)
        if k == 0:
            return
        C = self.capacity
        src, env, act = block.src_step, block.env, block.action
        rew, nxt, done = block.reward, block.next_step, block.done
        if k > C:
            src, env, act, rew, nxt, done = (v[-C:] for v in (src, env, act, rew, nxt, done))
            k = C
        slots = (self.pos + np.arange(k)) % C
        # states into slots [0, C), next states into [C, 2C): one pass
        self._store_indexed(block.obs_list, block.start_obs, np.concatenate([src, nxt]), np.concatenate([env, env]),
                            np.concatenate([slots, slots + C]), block.maker_side)
        dev = self.device
        # slots | actions | done flags | reward bit patterns in ONE host->device copy
        stage = np.empty((4, k), dtype=np.int64)
        stage[0], stage[1], stage[2] = slots, act, done
        stage[3] = np.ascontiguousarray(rew, dtype=np.float32).view(np.int32)
        sd = torch.from_numpy(stage).to(dev)
        st = sd[0]
        self.action.index_copy_(0, st, sd[1])
        self.done.index_copy_(0, st, sd[2].bool())
        self.reward.index_copy_(0, st, sd[3].int().view(torch.float32))
        self.pos = int((self.pos + k) % C)
        self.size = min(C, self.size + k)
        if self.prioritized:
            self._tree_update_td(st, None)         # new transitions enter at the running maximum priority

    # ---- sampling ----------------------------------------------------------------------------------------
    def _build_batch(self, slots_dev: torch.Tensor, slots_host: np.ndarray, starts=None) -> Batch:
        L = _lib.lib()
        dev = self.device
        k = len(slots_host)
        node_off = np.zeros(k + 1, dtype=np.int64)
        edge_off = np.zeros(k + 1, dtype=np.int64)
        np.cumsum(self.n_nodes[slots_host], out=node_off[1:])
        np.cumsum(self.n_edges[slots_host], out=edge_off[1:])
        N, E = int(node_off[-1]), int(edge_off[-1])
        offs = torch.from_numpy(np.concatenate([node_off, edge_off]).astype(np.int32)).to(dev, non_blocking=True)
        x = torch.empty((N, 3), dtype=torch.float32, device=dev)
        backmap = torch.empty(N, dtype=torch.long, device=dev)
        batch_vec = torch.empty(N, dtype=torch.long, device=dev)
        edge_local = torch.empty((2, max(E, 1)), dtype=torch.long, device=dev)
        edge_global = torch.empty((2, max(E, 1)), dtype=torch.long, device=dev)
        rowptr = torch.empty(N + 1, dtype=torch.int32, device=dev)
        col = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
        invdeg = torch.empty(max(N, 1), dtype=torch.float32, device=dev)
        idx32 = slots_dev.to(torch.int32)
        _lib.check(L.hexgnn_states_observe(self.hex_size, k, self.adj.data_ptr(), self.alive.data_ptr(),
                                           self.side.data_ptr(), idx32.data_ptr(), offs.data_ptr(),
                                           offs[k + 1:].data_ptr(), E, x.data_ptr(), backmap.data_ptr(),
                                           edge_local.data_ptr(), edge_global.data_ptr(), rowptr.data_ptr(),
                                           col.data_ptr(), invdeg.data_ptr(), batch_vec.data_ptr(), ops._stream()),
                   "hexgnn_states_observe")
        b = Batch()
        b.x, b.edge_index, b.batch = x, edge_global[:, :E], batch_vec
        b.ptr = torch.from_numpy(node_off).to(dev, non_blocking=True)
        b._num_graphs = k
        b.x._hex_max_nodes = int(self.n_nodes[slots_host].max()) if k else 0
        b.x._hex_hint_version = b.x._version
        gs = ops.GraphStructure.from_csr(N, E, rowptr, col, invdeg)
        if starts is not None and len(starts) - 1 <= self._max_blocks:
            gs.blocks = (torch.tensor(starts, dtype=torch.int32).to(dev, non_blocking=True), len(starts) - 1)
        b.edge_index._hex_csr = gs
        return b

    def sample(self, batch_size: int, beta: Optional[float] = None, generator: Optional[torch.Generator] = None):
        """Returns ``(indices, weights, state, next_state, action, reward, done)`` (weights all one when the buffer is
        not prioritized), states as device ``Batch`` objects.  One device->host copy of the sampled indices is needed
        to size the batches (the graph sizes live on the host)."""
        return self.sample_end(self.sample_begin(batch_size, beta, generator))

    def sample_begin(self, batch_size: int, beta: Optional[float] = None, generator: Optional[torch.Generator] = None):
        """First half of ``sample``: draws the indices on the device (stream-ordered behind every priority update issued
        so far) and starts their copy to pinned host memory; returns a handle for ``sample_end``.  A loop with several
        buffers can begin the next buffer's draw before it waits for this one, so that the host builds one batch while
        the GPU still runs the previous update (examples/selfplay_train.py)."""
        if self.size == 0:
            raise ValueError("empty buffer")
        dev = self.device
        if self.prioritized:
            u = torch.rand(batch_size, dtype=torch.float64, device=dev, generator=generator)
            idx = torch.empty(batch_size, dtype=torch.int32, device=dev)
            w = torch.empty(batch_size, dtype=torch.float32, device=dev)
            _lib.check(_lib.lib().hexgnn_per_sample(self.cap2, self.size, batch_size, float(beta if beta is not None else 0.4),
                                                    u.data_ptr(), self.sum_tree.data_ptr(), self.min_tree.data_ptr(),
                                                    idx.data_ptr(), w.data_ptr(), ops._stream()), "hexgnn_per_sample")
            idx = idx.long()
        else:
            idx = torch.randint(0, self.size, (batch_size,), device=dev, generator=generator)
            w = torch.ones(batch_size, dtype=torch.float32, device=dev)
        host = torch.empty(batch_size, dtype=torch.int64, pin_memory=True)
        host.copy_(idx, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return (idx, w, host, ev, (self.pos, self.size))

    def sample_end(self, pending):
        idx, w, host_t, ev, stamp = pending
        if stamp != (self.pos, self.size):
            raise RuntimeError("transitions were stored between sample_begin and sample_end: the drawn slots may have been "
                               "overwritten")
        ev.synchronize()
        host = host_t.numpy().copy()
        batch_size = len(host)
        st_blocks = nx_blocks = None
        if self.pack_blocks and batch_size:
            from .data import blocks_for_order, pack_order
            self._max_blocks = ops.stack_block_budget(self.device)      # (a host-side query; changes with GradSync.enable_overlap)
            order, st_blocks = pack_order(self.n_nodes[host], max_blocks=self._max_blocks)
            if st_blocks is not None:
                perm = np.asarray(order, dtype=np.int64)
                host = host[perm]
                pd = torch.from_numpy(perm).to(self.device, non_blocking=True)
                idx, w = idx[pd], w[pd]
                nx_blocks = blocks_for_order(self.n_nodes[host + self.capacity])      # (same order: fewer nodes per graph)
        state = self._build_batch(idx, host, st_blocks)
        nxt = self._build_batch(idx + self.capacity, host + self.capacity, nx_blocks)
        # all stored states of one buffer share the mover's side (maker / breaker buffers are separate)
        state.x._hex_is_maker = bool(self.side_host[int(host[0])]) if batch_size else True
        state.x._hex_hint_version = state.x._version
        nxt.x._hex_is_maker = bool(self.side_host[int(host[0]) + self.capacity]) if batch_size else True
        nxt.x._hex_hint_version = nxt.x._version
        return idx, w, state, nxt, self.action[idx], self.reward[idx], self.done[idx]

    def update_priorities(self, indices: torch.Tensor, td_errors: torch.Tensor) -> None:
        """priority_i = |td_i| + eps (the running maximum is raised to the largest), leaf = priority^alpha: one launch."""
        if not self.prioritized:
            return
        idx = indices.to(device=self.device)
        if idx.dtype not in (torch.int32, torch.int64):
            idx = idx.long()
        idx = idx.contiguous().flatten()
        td = td_errors.detach().to(device=self.device, dtype=torch.float32).contiguous().flatten()
        if td.numel() != idx.numel():
            raise ValueError("indices and td_errors differ in length")
        self._tree_update_td(idx, td)

    def _tree_update_td(self, idx: torch.Tensor, td: Optional[torch.Tensor]):
        _lib.check(_lib.lib().hexgnn_per_update_td(self.cap2, int(idx.numel()), idx.data_ptr(),
                                                   64 if idx.dtype == torch.int64 else 32,
                                                   td.data_ptr() if td is not None else None, self.alpha, self.eps,
                                                   self.max_priority.data_ptr(), self.sum_tree.data_ptr(),
                                                   self.min_tree.data_ptr(), ops._stream()), "hexgnn_per_update_td")
