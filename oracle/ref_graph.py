"""ctypes access to oracle/_ref/libhexgraph_ref.so: the REFERENCE's own CSR graph container
(cpp_hex/hex_graph_game/graph.cpp, compiled in place by oracle/Makefile).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(_HERE, "_ref", "libhexgraph_ref.so")


def available() -> bool:
    return os.path.exists(SO)


class RefGraph:
    def __init__(self, num_vertices: int):
        L = C.CDLL(SO)
        L.refgraph_new.restype = C.c_void_p
        L.refgraph_new.argtypes = [C.c_int]
        L.refgraph_free.argtypes = [C.c_void_p]
        for n in ("refgraph_add_edge", "refgraph_delete_edge", "refgraph_edge_exists"):
            getattr(L, n).restype = C.c_int
            getattr(L, n).argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.refgraph_clear_vertex.argtypes = [C.c_void_p, C.c_int]
        for n in ("refgraph_num_vertices", "refgraph_num_directed_edges"):
            getattr(L, n).restype = C.c_int
            getattr(L, n).argtypes = [C.c_void_p]
        L.refgraph_dump.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        self._L = L
        self._h = L.refgraph_new(num_vertices)

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.refgraph_free(self._h)
            self._h = None

    def add_edge(self, a, b): return bool(self._L.refgraph_add_edge(self._h, a, b))
    def delete_edge(self, a, b): return bool(self._L.refgraph_delete_edge(self._h, a, b))
    def edge_exists(self, a, b): return bool(self._L.refgraph_edge_exists(self._h, a, b))
    def clear_vertex(self, v): self._L.refgraph_clear_vertex(self._h, v)

    def dump(self):
        nv = self._L.refgraph_num_vertices(self._h)
        ne = self._L.refgraph_num_directed_edges(self._h)
        s = np.empty(max(ne, 1), dtype=np.int32)
        t = np.empty(max(ne, 1), dtype=np.int32)
        es = np.empty(nv + 1, dtype=np.int32)
        self._L.refgraph_dump(self._h, s.ctypes.data, t.ctypes.data, es.ctypes.data)
        return s[:ne], t[:ne], es


def start_graph_via_reference_container(size: int):
    """Replay the insertion sequence of Node_switching_game::reset_graph
    (cpp_hex/hex_graph_game/shannon_node_switching_game.cpp:139-156, SINGLE graph) through the reference's
    Graph::add_edge and return its CSR (sources, targets, edge_starts)."""
    n, sq = size, size * size
    g = RefGraph(sq + 2)
    for i in range(sq):
        j = i + 2
        if i < n:
            g.add_edge(j, 0)
        if i // n == n - 1:
            g.add_edge(j, 1)
        if i % n > 0 and n <= i <= sq - n:
            g.add_edge(j, j - 1)
        if i >= n:
            g.add_edge(j, j - n)
            if i % n != n - 1:
                g.add_edge(j, j + 1 - n)
    return g.dump()
