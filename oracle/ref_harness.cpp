// Harness around the REFERENCE's own CSR graph container, compiled in place from
// /root/reference/cpp_hex/hex_graph_game/graph.cpp (never copied into this repo).
// TEST INFRASTRUCTURE ONLY (oracle/__init__.py).  Built by oracle/Makefile into
// oracle/_ref/libhexgraph_ref.so, in this container only; the .so travels to the GPU box.
//
// Only graph.cpp / hex_board_game.cpp build from their own sources here.  The game logic
// (shannon_node_switching_game.cpp) includes util.h -> <blaze/Math.h> + CrazyAra's nn_api.h,
// which the image lacks; building it would need stand-in headers, so it is treated as
// unbuildable (DESIGN.md section "Oracle").  What this harness pins is therefore the
// container semantics the native twin is built on: sorted adjacency rows, duplicate-free
// add_edge, delete_edge, clear_vertex, and the start-graph edge list when the insertion
// sequence of Node_switching_game::reset_graph (shannon_node_switching_game.cpp:139-156) is
// replayed by the caller through refgraph_add_edge.
#include <cstdint>
#include "graph.h"

extern "C" {

void* refgraph_new(int num_vertices) { return new Graph(num_vertices); }
void refgraph_free(void* g) { delete static_cast<Graph*>(g); }
int refgraph_add_edge(void* g, int a, int b) { return static_cast<Graph*>(g)->add_edge(a, b) ? 1 : 0; }
int refgraph_delete_edge(void* g, int a, int b) { return static_cast<Graph*>(g)->delete_edge(a, b) ? 1 : 0; }
int refgraph_edge_exists(void* g, int a, int b) { return static_cast<Graph*>(g)->edge_exists(a, b) ? 1 : 0; }
void refgraph_clear_vertex(void* g, int v) { static_cast<Graph*>(g)->clear_vertex(v); }
int refgraph_num_vertices(void* g) { return static_cast<Graph*>(g)->num_vertices; }
int refgraph_num_directed_edges(void* g) { return (int)static_cast<Graph*>(g)->sources.size(); }
// CSR dump: sources[e], targets[e] (e < num_directed_edges), edge_starts[num_vertices+1]
void refgraph_dump(void* gp, int* sources, int* targets, int* edge_starts) {
    Graph* g = static_cast<Graph*>(gp);
    for (size_t i = 0; i < g->sources.size(); ++i) { sources[i] = g->sources[i]; targets[i] = g->targets[i]; }
    for (size_t i = 0; i < g->edge_starts.size(); ++i) edge_starts[i] = g->edge_starts[i];
}

}  // extern "C"
