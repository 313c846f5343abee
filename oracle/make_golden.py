"""Generates tests/golden/*.npz.  Run in THIS container only (python oracle/make_golden.py):

* start_graph_csr.npz  -- CSR of the Hex-5..13 start graphs produced by the REFERENCE's own Graph container
                          (oracle/_ref, compiled from /root/reference/cpp_hex/hex_graph_game/graph.cpp) when
                          the insertion sequence of reset_graph is replayed through it.
* playouts.npz         -- seeded random playouts of the env oracle (canonical ascending order): for each game the
                          move list (vertex ids), who moved first, the winner, the number of moves and, after
                          every move with dead/captured removal, (n_alive, n_edges, fnv1a64(adjacency bit matrix)).
                          Build-owned fixtures: they pin the canonical order for the HIP builder and for
                          regressions of the oracle itself (the reference's two implementations disagree on
                          reduced graphs).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import env_ref, ref_graph  # noqa: E402


def fnv1a64(b: bytes) -> int:
    h = 0xcbf29ce484222325
    for x in b:
        h = ((h ^ x) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return h


def state_digest(game):
    adj, alive = game.dump()
    return int(alive.sum()), game.num_edges(), fnv1a64(adj.tobytes() + alive.tobytes())


def main():
    out = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out, exist_ok=True)
    csr = {}
    for n in range(5, 14):
        s, t, es = ref_graph.start_graph_via_reference_container(n)
        csr["s%d" % n], csr["t%d" % n], csr["es%d" % n] = s, t, es
    np.savez_compressed(os.path.join(out, "start_graph_csr.npz"), **csr)

    env_ref.build()
    rec = {}
    gid = 0
    for size, games in ((5, 12), (7, 12), (11, 16)):
        for k in range(games):
            rng = np.random.Generator(np.random.PCG64(9000 + gid))
            game = env_ref.RefGame(size)
            maker_first = (k % 2 == 0)
            game.maker_turn = maker_first
            moves, digests = [], []
            while game.who_won() is None:
                acts = game.get_actions()
                mv = int(acts[rng.integers(0, len(acts))])
                game.make_move(mv, remove_dead_and_captured=True)
                moves.append(mv)
                digests.append(state_digest(game))
            w = {"m": 0, "b": 1}[game.who_won()]
            rec["g%d_meta" % gid] = np.array([size, int(maker_first), w, len(moves)], dtype=np.int64)
            rec["g%d_moves" % gid] = np.array(moves, dtype=np.int64)
            rec["g%d_digest" % gid] = np.array(digests, dtype=np.uint64)
            gid += 1
    rec["num_games"] = np.array([gid])
    np.savez_compressed(os.path.join(out, "playouts.npz"), **rec)
    print("wrote", out)


if __name__ == "__main__":
    main()
