"""ctypes driver for the C env oracle (``oracle/env_ref.c``) + the python-level
``Env_manager`` semantics restated on top of it.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.

``RefGame``        one game (graph_game/graph_tools_games.py:20-29 ``Hex_game``)
``RefEnvManager``  graph_game/multi_env_manager.py:16-165, GNN observation mode only
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import time
from typing import List

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build() -> str:
    """Compile oracle/env_ref.c -> oracle/libhexref.so (idempotent)."""
    so = os.path.join(_HERE, "libhexref.so")
    src = os.path.join(_HERE, "env_ref.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-Wall", "-o", so, src])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libhexref.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.hexref_new.restype = C.c_void_p
        L.hexref_new.argtypes = [C.c_int]
        L.hexref_copy.restype = C.c_void_p
        L.hexref_copy.argtypes = [C.c_void_p]
        for name in ("hexref_free", "hexref_reset"):
            getattr(L, name).restype = None
            getattr(L, name).argtypes = [C.c_void_p]
        L.hexref_set_maker_turn.restype = None
        L.hexref_set_maker_turn.argtypes = [C.c_void_p, C.c_int]
        for name in ("hexref_maker_turn", "hexref_total_num_moves", "hexref_num_vertices", "hexref_who_won",
                     "hexref_num_edges", "hexref_words", "hexref_nv"):
            getattr(L, name).restype = C.c_int
            getattr(L, name).argtypes = [C.c_void_p]
        L.hexref_make_move.restype = C.c_int
        L.hexref_make_move.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.hexref_get_response.restype = C.c_int
        L.hexref_get_response.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.hexref_get_actions.restype = C.c_int
        L.hexref_get_actions.argtypes = [C.c_void_p, C.c_void_p]
        L.hexref_observe.restype = C.c_int
        L.hexref_observe.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.hexref_dump.restype = None
        L.hexref_dump.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _LIB = L
    return _LIB


class RefGame:
    """Hex_game(size) on the C oracle.  Vertex ids: 0,1 terminals, i+2 = board cell i."""

    def __init__(self, size: int, _handle=None):
        self.size = size
        self._L = lib()
        self._h = _handle if _handle is not None else self._L.hexref_new(size)
        if not self._h:
            raise ValueError("unsupported board size %d" % size)
        self.creation_time = time.perf_counter()

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.hexref_free(self._h)
            self._h = None

    def copy(self) -> "RefGame":
        return RefGame(self.size, self._L.hexref_copy(self._h))

    @property
    def nv(self) -> int:
        return self.size * self.size + 2

    @property
    def maker_turn(self) -> bool:
        return bool(self._L.hexref_maker_turn(self._h))

    @maker_turn.setter
    def maker_turn(self, m: bool):
        self._L.hexref_set_maker_turn(self._h, int(bool(m)))

    @property
    def onturn(self) -> str:
        return "m" if self.maker_turn else "b"

    @property
    def not_onturn(self) -> str:
        return "b" if self.maker_turn else "m"

    @property
    def total_num_moves(self) -> int:
        return self._L.hexref_total_num_moves(self._h)

    def make_move(self, vertex: int, remove_dead_and_captured: bool = False):
        if self._L.hexref_make_move(self._h, int(vertex), int(remove_dead_and_captured)) != 0:
            raise ValueError("illegal move %d" % vertex)

    def get_response(self, move: int, for_maker: bool):
        r = self._L.hexref_get_response(self._h, int(move), int(bool(for_maker)))
        return None if r < 0 else r

    def who_won(self):
        w = self._L.hexref_who_won(self._h)
        return {0: "m", 1: "b", -1: None}[w]

    def get_actions(self) -> np.ndarray:
        buf = np.empty(self.nv, dtype=np.int32)
        k = self._L.hexref_get_actions(self._h, buf.ctypes.data)
        return buf[:k].astype(np.int64)

    def num_vertices(self) -> int:
        return self._L.hexref_num_vertices(self._h)

    def num_edges(self) -> int:
        return self._L.hexref_num_edges(self._h)

    def observe(self):
        """(x [n,3] f32, edge_index [2,2E] i64, backmap [n] i64) as numpy arrays."""
        n = self.num_vertices()
        e2 = 2 * self.num_edges()
        x = np.empty((n, 3), dtype=np.float32)
        ei = np.empty((2, e2), dtype=np.int64)
        bm = np.empty(n, dtype=np.int64)
        e2c = C.c_int(0)
        n2 = self._L.hexref_observe(self._h, x.ctypes.data, ei.ctypes.data, bm.ctypes.data, C.byref(e2c))
        assert n2 == n and e2c.value == e2
        return x, ei, bm

    def dump(self):
        """(adj [nv, words] u64 bit matrix, alive [nv] u8)."""
        words = self._L.hexref_words(self._h)
        adj = np.empty((self.nv, words), dtype=np.uint64)
        alive = np.empty(self.nv, dtype=np.uint8)
        self._L.hexref_dump(self._h, adj.ctypes.data, alive.ctypes.data)
        return adj, alive


class RefObs:
    """Minimal stand-in for torch_geometric.data.Data used by the oracle manager (numpy payload)."""

    def __init__(self, x, edge_index, backmap=None):
        self.x = x
        self.edge_index = edge_index
        if backmap is not None:
            self.backmap = backmap


class RefEnvManager:
    """graph_game/multi_env_manager.py:16-165 on RefGame (cnn_rep=False path)."""

    def __init__(self, num_envs, hex_size, gamma=1, n_steps=[1], prune_exploratories=True):
        self.num_envs = num_envs
        self.gamma = gamma
        self.global_onturn = "m"
        self.n_steps = n_steps
        self.prune_exploratories = prune_exploratories
        self.change_hex_size(hex_size)

    def change_hex_size(self, new_size):
        self.hex_size = new_size
        self.global_onturn = "m"
        self.envs = [RefGame(new_size) for _ in range(self.num_envs)]
        self.base_game = RefGame(new_size)

    @staticmethod
    def _obs(game: RefGame) -> RefObs:
        return RefObs(*game.observe())

    @property
    def starting_obs(self):
        return self._obs(self.base_game)

    def observe(self) -> List[RefObs]:
        return [self._obs(e) for e in self.envs]

    @staticmethod
    def validate_actions(states, actions):
        return [int(s.backmap[a]) for s, a in zip(states, actions)]

    def get_valid_actions(self):
        return [e.get_actions() for e in self.envs]

    def step(self, actions):
        rewards = np.zeros(self.num_envs, dtype=float)
        dones = np.zeros(self.num_envs, dtype=bool)
        infos = [{} for _ in range(self.num_envs)]
        for i, (act, env) in enumerate(zip(actions, self.envs)):
            env.make_move(int(act), remove_dead_and_captured=True)
            winner = env.who_won()
            if winner is not None:
                dones[i] = True
                n = env.total_num_moves
                infos[i]["episode_metrics"] = {
                    "return": 1 if winner == "m" else -1,
                    "discounted_return": float(1 * self.gamma ** n if winner == "m" else -1 * self.gamma ** n),
                    "length": n,
                    "time": time.perf_counter() - env.creation_time,
                }
                rewards[i] = 1 if winner == env.not_onturn else -1
                self.envs[i] = RefGame(self.hex_size)
                self.envs[i].maker_turn = self.global_onturn == "b"
        self.global_onturn = "m" if self.global_onturn == "b" else "b"
        assert self.global_onturn == self.envs[0].onturn
        return self.observe(), rewards, dones, infos

    def reset(self):
        self.global_onturn = "m"
        self.envs = [RefGame(self.hex_size) for _ in range(self.num_envs)]
        return self.observe()

    def get_transitions(self, starting_states, state_history, action_history, reward_history, done_history,
                        exploratories_history):
        """multi_env_manager.py:113-165.  Returns (maker_transitions, breaker_transitions)."""
        maker, breaker = [], []
        sh = list(state_history)
        sh.insert(0, starting_states)
        for i in range(len(action_history)):
            start_state = sh[i]
            action = action_history[i]
            transits = maker if start_state[0].x[0, 2] == 1 else breaker
            for n_step in self.n_steps:
                if len(sh) > i + 2 * n_step:
                    for k in range(len(start_state)):
                        if hasattr(start_state[k], "backmap"):
                            del start_state[k].backmap
                        assert action[k] < len(start_state[k].x)
                        reward = 0
                        for j in range(i, i + 2 * n_step):
                            reward += reward_history[j][k] * ((-((j - i) % 2)) * 2 + 1) * (self.gamma ** ((j - i) // 2))
                            if done_history[j][k]:
                                sobs = self.starting_obs
                                del sobs.backmap
                                sobs.x[:, 2] = start_state[k].x[0, 2]
                                transits.append((start_state[k], action[k], reward, sobs, True))
                                break
                            if self.prune_exploratories and j > i and exploratories_history[j][k]:
                                break
                        else:
                            nxt = sh[i + 2 * n_step][k]
                            if hasattr(nxt, "backmap"):
                                del nxt.backmap
                            transits.append((start_state[k], action[k], reward, nxt, False))
        return maker, breaker


# ---------------------------------------------------------------------------------------------
# synthetic batches of SURVEY section 8(d): D0 (start positions), D1 (seeded random playouts), MIX
# ---------------------------------------------------------------------------------------------

def random_position(size: int, g: int, want_maker_turn: bool) -> RefGame:
    """D1: start position after k ~ U{0..floor(n^2/2)} random legal moves with dead/captured removal,
    unfinished games only, RNG = PCG64(1234+g); k's parity fixed so that the side to move is uniform."""
    rng = np.random.Generator(np.random.PCG64(1234 + g))
    while True:
        k = int(rng.integers(0, size * size // 2 + 1))
        if (k % 2 == 0) != want_maker_turn:
            k = k + 1 if k == 0 else k - 1
        game = RefGame(size)
        ok = True
        for _ in range(k):
            acts = game.get_actions()
            game.make_move(int(acts[rng.integers(0, len(acts))]), remove_dead_and_captured=True)
            if game.who_won() is not None:
                ok = False
                break
        if ok and game.who_won() is None and len(game.get_actions()) > 0:
            return game


def make_batch(kind: str, sizes, maker_turn: bool = True):
    """Collate graphs into (x [N,3], edge_index [2,E], batch [N], ptr [B+1]) numpy arrays.

    kind 'D0': start positions; 'D1': random_position(size, g).  ``sizes`` = one board size per graph.
    Collation follows torch_geometric Batch.from_data_list / util.cpp:22-41 (node offset per graph)."""
    xs, eis, batch, ptr = [], [], [], [0]
    off = 0
    for g, size in enumerate(sizes):
        if kind == "D0":
            game = RefGame(size)
            game.maker_turn = maker_turn
        else:
            game = random_position(size, g, maker_turn)
        x, ei, _ = game.observe()
        xs.append(x)
        eis.append(ei + off)
        batch.append(np.full(x.shape[0], g, dtype=np.int64))
        off += x.shape[0]
        ptr.append(off)
    return (np.concatenate(xs, 0), np.concatenate(eis, 1), np.concatenate(batch), np.asarray(ptr, dtype=np.int64))
