/* CPU restatement of the reference's Hex / Shannon node-switching game logic.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): never linked into or
 * called from the product library.  Plain C, built by oracle/Makefile into
 * oracle/libhexref.so and driven through ctypes by oracle/env_ref.py.
 *
 * Follows the PYTHON implementation (the one the RainbowDQN path runs):
 *   start graph      graph_game/hex_board_game.py:214-233  (graph_from_board, redgraph=True,
 *                    no_worthless_edges=True); native twin
 *                    cpp_hex/hex_graph_game/shannon_node_switching_game.cpp:120-157
 *   make_move        graph_game/shannon_node_switching_game.py:80-116
 *   fix terminals    graph_game/shannon_node_switching_game.py:57-65
 *   dead_and_captured graph_game/shannon_node_switching_game.py:119-196
 *   who_won          graph_game/shannon_node_switching_game.py:199-205
 *   get_actions      graph_game/shannon_node_switching_game.py:47-48
 *   is_fully_connected / double_loop_iterator   graph_game/utils.py:33-38,104-116
 *   observation      GN0/util/convert_graph.py:77-85,99-122 (old_style=True)
 *
 * Canonical order.  The python code iterates python `set`s and graph-tool
 * neighbour lists, whose order is an implementation accident (CPython hash
 * order / edge insertion order) and differs from the C++ twin (std::set,
 * sorted CSR).  This restatement fixes ONE order: every set and every
 * neighbour list is visited in ASCENDING vertex id, the checks per vertex run
 * in python's sequence (dead, then maker-capture, then breaker-capture), the
 * maker-capture twin is the first hit of chain([one_neighbor],
 * neighbours(one_neighbor)) with one_neighbor = smallest neighbour.  The HIP
 * builder implements the same order and is compared bit for bit.
 *
 * Vertices keep their original ids (0,1 = terminals, i+2 = board cell i); a
 * removed vertex has alive=0 and an empty adjacency row (a graph-tool
 * GraphView hides every edge of a filtered vertex, so clearing is equivalent).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MAXW 10 /* 640 vertices: up to Hex-25 (627) */

typedef struct {
    int size, nv, words;
    int maker_turn;       /* view.gp["m"] */
    int total_num_moves;  /* Node_switching_game.total_num_moves */
    uint8_t* alive;       /* vp.f */
    uint64_t* adj;        /* nv x words, symmetric */
    int* resp_maker;      /* response_set_maker  (py:31,160-161), -1 = absent */
    int* resp_breaker;    /* response_set_breaker (py:32,182-183) */
} game_t;

typedef struct { uint64_t w[MAXW]; } vset;

static inline uint64_t* row(const game_t* g, int v) { return g->adj + (size_t)v * g->words; }
static inline int has_edge(const game_t* g, int a, int b) { return (int)((row(g, a)[b >> 6] >> (b & 63)) & 1u); }
static inline void add_edge(game_t* g, int a, int b) {
    row(g, a)[b >> 6] |= 1ull << (b & 63);
    row(g, b)[a >> 6] |= 1ull << (a & 63);
}
static inline void del_edge(game_t* g, int a, int b) {
    row(g, a)[b >> 6] &= ~(1ull << (b & 63));
    row(g, b)[a >> 6] &= ~(1ull << (a & 63));
}
static inline void vs_clear(vset* s) { memset(s->w, 0, sizeof(s->w)); }
static inline void vs_add(vset* s, int v) { s->w[v >> 6] |= 1ull << (v & 63); }
static inline int vs_empty(const vset* s, int words) {
    for (int i = 0; i < words; ++i) if (s->w[i]) return 0;
    return 1;
}
/* ascending list of set bits */
static int vs_list(const uint64_t* w, int words, int* out) {
    int k = 0;
    for (int i = 0; i < words; ++i) {
        uint64_t m = w[i];
        while (m) { int b = __builtin_ctzll(m); out[k++] = i * 64 + b; m &= m - 1; }
    }
    return k;
}
static inline int neighbors(const game_t* g, int v, int* out) { return vs_list(row(g, v), g->words, out); }

game_t* hexref_new(int size) {
    game_t* g = (game_t*)calloc(1, sizeof(game_t));
    g->size = size;
    g->nv = size * size + 2;
    g->words = (g->nv + 63) / 64;
    if (g->words > MAXW) { free(g); return NULL; }
    g->alive = (uint8_t*)malloc((size_t)g->nv);
    g->adj = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)g->nv * g->words);
    g->resp_maker = (int*)malloc(sizeof(int) * (size_t)g->nv);
    g->resp_breaker = (int*)malloc(sizeof(int) * (size_t)g->nv);
    extern void hexref_reset(game_t*);
    hexref_reset(g);
    return g;
}

void hexref_free(game_t* g) { if (g) { free(g->alive); free(g->adj); free(g->resp_maker); free(g->resp_breaker); free(g); } }

/* hex_board_game.py:214-233 */
void hexref_reset(game_t* g) {
    const int n = g->size, sq = n * n;
    memset(g->adj, 0, sizeof(uint64_t) * (size_t)g->nv * g->words);
    memset(g->alive, 1, (size_t)g->nv);
    for (int v = 0; v < g->nv; ++v) g->resp_maker[v] = g->resp_breaker[v] = -1;
    for (int i = 0; i < sq; ++i) {
        const int v = i + 2;
        if (i < n) add_edge(g, v, 0);
        if (i / n == n - 1) add_edge(g, v, 1);
        if (i % n > 0 && n <= i && i <= sq - n) add_edge(g, v, v - 1);
        if (i >= n) {
            add_edge(g, v, v - n);
            if (i % n != n - 1) add_edge(g, v, v - n + 1);
        }
    }
    g->maker_turn = 1;
    g->total_num_moves = 0;
}

game_t* hexref_copy(const game_t* s) {
    game_t* g = (game_t*)malloc(sizeof(game_t));
    *g = *s;
    g->alive = (uint8_t*)malloc((size_t)s->nv);
    g->adj = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)s->nv * s->words);
    memcpy(g->alive, s->alive, (size_t)s->nv);
    memcpy(g->adj, s->adj, sizeof(uint64_t) * (size_t)s->nv * s->words);
    g->resp_maker = (int*)malloc(sizeof(int) * (size_t)s->nv);
    g->resp_breaker = (int*)malloc(sizeof(int) * (size_t)s->nv);
    memcpy(g->resp_maker, s->resp_maker, sizeof(int) * (size_t)s->nv);
    memcpy(g->resp_breaker, s->resp_breaker, sizeof(int) * (size_t)s->nv);
    return g;
}

void hexref_set_maker_turn(game_t* g, int m) { g->maker_turn = m ? 1 : 0; }
int hexref_maker_turn(const game_t* g) { return g->maker_turn; }
int hexref_total_num_moves(const game_t* g) { return g->total_num_moves; }
int hexref_num_vertices(const game_t* g) { int k = 0; for (int v = 0; v < g->nv; ++v) k += g->alive[v]; return k; }

/* shannon_node_switching_game.py:57-65 */
static void fix_terminal_connections(game_t* g, int terminal, vset* change) {
    int nb[MAXW * 64];
    const int k = neighbors(g, terminal, nb);
    for (int i = 0; i < k; ++i)
        for (int j = i + 1; j < k; ++j)
            if (has_edge(g, nb[i], nb[j])) {
                del_edge(g, nb[i], nb[j]);
                vs_add(change, nb[i]);
                vs_add(change, nb[j]);
            }
}

static void dead_and_captured(game_t* g, const vset* consider_in);

/* shannon_node_switching_game.py:80-116.  force: -1 = player on turn, 0 = 'b', 1 = 'm'. */
static void make_move(game_t* g, int v, int force, int remove_dc, vset* change_out) {
    int makerturn;
    vset change; vs_clear(&change);
    if (force < 0) { g->total_num_moves++; makerturn = g->maker_turn; }
    else makerturn = force;
    if (makerturn) {
        int nb[MAXW * 64];
        const int k = neighbors(g, v, nb);
        int have_to_fix = -1;
        for (int i = 0; i < k; ++i)
            for (int j = i + 1; j < k; ++j) {
                const int v1 = nb[i], v2 = nb[j];
                if (v1 < 2) have_to_fix = v1;
                else if (v2 < 2) have_to_fix = v2;
                if (!((has_edge(g, v1, 0) && has_edge(g, v2, 0)) || (has_edge(g, v1, 1) && has_edge(g, v2, 1))))
                    add_edge(g, v1, v2);
            }
        if (have_to_fix >= 0) fix_terminal_connections(g, have_to_fix, &change);
    }
    /* vp.f[v] = False; the consider set below is read from the still present row of v */
    vset consider = change;
    {
        const uint64_t* r = row(g, v);
        for (int i = 0; i < g->words; ++i) consider.w[i] |= r[i];
    }
    g->alive[v] = 0;
    {   /* hide every edge of v */
        int nb[MAXW * 64];
        const int k = neighbors(g, v, nb);
        for (int i = 0; i < k; ++i) del_edge(g, v, nb[i]);
    }
    if (force < 0) g->maker_turn = !g->maker_turn;
    if (remove_dc) dead_and_captured(g, &consider);
    if (change_out) *change_out = change;
}

/* graph_game/utils.py:104-116 over an explicit vertex list */
static int fully_connected_list(const game_t* g, const int* vs, int k) {
    for (int i = 0; i < k; ++i)
        for (int j = i + 1; j < k; ++j)
            if (!has_edge(g, vs[i], vs[j])) return 0;
    return 1;
}
static int fully_connected_set(const game_t* g, const vset* s) {
    int vs[MAXW * 64];
    const int k = vs_list(s->w, g->words, vs);
    return fully_connected_list(g, vs, k);
}

/* shannon_node_switching_game.py:119-196, iterate=True */
static void dead_and_captured(game_t* g, const vset* consider_in) {
    vset consider = *consider_in;
    const int W = g->words;
    while (!vs_empty(&consider, W)) {
        vset big; vs_clear(&big);
        int nodes[MAXW * 64];
        const int nc = vs_list(consider.w, W, nodes);
        for (int ci = 0; ci < nc; ++ci) {
            const int node = nodes[ci];
            if (!g->alive[node] || node < 2) continue;
            vset neighset;
            memcpy(neighset.w, row(g, node), sizeof(uint64_t) * W);
            for (int i = W; i < MAXW; ++i) neighset.w[i] = 0;
            if (fully_connected_set(g, &neighset)) {            /* dead */
                for (int i = 0; i < W; ++i) big.w[i] |= neighset.w[i];
                make_move(g, node, 0, 0, NULL);
                continue;
            }
            int nb[MAXW * 64];
            const int k = vs_list(neighset.w, W, nb);
            /* maker capture: chain([one_neighbor], neighbours(one_neighbor)) */
            int made_move = 0;
            {
                const int one = nb[0];
                int cand[MAXW * 64 + 1];
                cand[0] = one;
                const int kc = 1 + neighbors(g, one, cand + 1);
                for (int q = 0; q < kc && !made_move; ++q) {
                    const int nbr = cand[q];
                    if (nbr < 2 || nbr == node) continue;
                    vset without_me, without_him;
                    int same = 1;
                    for (int i = 0; i < MAXW; ++i) {
                        without_me.w[i] = i < W ? row(g, nbr)[i] : 0;
                        without_him.w[i] = neighset.w[i];
                    }
                    without_me.w[node >> 6] &= ~(1ull << (node & 63));
                    without_him.w[nbr >> 6] &= ~(1ull << (nbr & 63));
                    for (int i = 0; i < W; ++i) if (without_me.w[i] != without_him.w[i]) { same = 0; break; }
                    if (same) {
                        vset change;
                        g->resp_maker[node] = nbr;
                        g->resp_maker[nbr] = node;
                        make_move(g, nbr, 0, 0, NULL);
                        make_move(g, node, 1, 0, &change);
                        for (int i = 0; i < W; ++i) big.w[i] |= without_me.w[i] | change.w[i];
                        made_move = 1;
                    }
                }
            }
            if (made_move) continue;
            /* breaker capture */
            for (int q = 0; q < k; ++q) {
                const int nbr = nb[q];
                if (nbr < 2) continue;
                vset without_me, without_him;
                for (int i = 0; i < MAXW; ++i) {
                    without_me.w[i] = i < W ? row(g, nbr)[i] : 0;
                    without_him.w[i] = neighset.w[i];
                }
                without_me.w[node >> 6] &= ~(1ull << (node & 63));
                without_him.w[nbr >> 6] &= ~(1ull << (nbr & 63));
                if (fully_connected_set(g, &without_me) && fully_connected_set(g, &without_him)) {
                    for (int i = 0; i < W; ++i) big.w[i] |= without_me.w[i] | without_him.w[i];
                    g->resp_breaker[node] = nbr;
                    g->resp_breaker[nbr] = node;
                    make_move(g, node, 0, 0, NULL);
                    make_move(g, nbr, 0, 0, NULL);
                    break;
                }
            }
        }
        consider = big;
    }
}

/* public move by the player on turn (Env_manager.step: make_move(int(act), remove_dead_and_captured=True)) */
int hexref_make_move(game_t* g, int vertex, int remove_dead_and_captured) {
    if (vertex < 2 || vertex >= g->nv || !g->alive[vertex]) return -1;
    make_move(g, vertex, -1, remove_dead_and_captured, NULL);
    return 0;
}

/* py:67-78: the stored reply to `move` for the given side, or -1 */
int hexref_get_response(const game_t* g, int move, int for_maker) {
    if (move < 0 || move >= g->nv) return -1;
    return for_maker ? g->resp_maker[move] : g->resp_breaker[move];
}

/* 0 = maker ('m'), 1 = breaker ('b'), -1 = undecided (None).  py:199-205 */
int hexref_who_won(const game_t* g) {
    if (has_edge(g, 0, 1)) return 0;
    uint8_t seen[MAXW * 64];
    int stack[MAXW * 64], sp = 0;
    memset(seen, 0, sizeof(seen));
    seen[0] = 1; stack[sp++] = 0;
    while (sp) {
        const int v = stack[--sp];
        int nb[MAXW * 64];
        const int k = neighbors(g, v, nb);
        for (int i = 0; i < k; ++i) {
            if (nb[i] == 1) return -1;
            if (!seen[nb[i]]) { seen[nb[i]] = 1; stack[sp++] = nb[i]; }
        }
    }
    return 1;
}

/* py:47-48: alive vertex ids except the terminals, ascending */
int hexref_get_actions(const game_t* g, int* out) {
    int k = 0;
    for (int v = 2; v < g->nv; ++v) if (g->alive[v]) out[k++] = v;
    return k;
}

int hexref_num_edges(const game_t* g) {
    int e = 0;
    for (int v = 0; v < g->nv; ++v)
        for (int i = 0; i < g->words; ++i) e += __builtin_popcountll(row(g, v)[i]);
    return e / 2;
}

/* convert_graph.py:77-85,99-122 (old_style=True, global_input_properties=[m]).
 * x: [n,3] f32 = (degree, is_terminal, maker_to_move); backmap: [n] rank -> vertex id;
 * edge_index: [2, 2E] int64, first the E edges (s > t, sorted by (s,t)) then the flipped copies.
 * Returns n; *e2 = 2E.  */
int hexref_observe(const game_t* g, float* x, int64_t* edge_index, int64_t* backmap, int* e2) {
    int rank[MAXW * 64];
    int n = 0;
    for (int v = 0; v < g->nv; ++v) if (g->alive[v]) { rank[v] = n; backmap[n] = v; ++n; } else rank[v] = -1;
    const int E = hexref_num_edges(g);
    int k = 0;
    for (int s = 0; s < g->nv; ++s) {
        if (!g->alive[s]) continue;
        int nb[MAXW * 64];
        const int d = neighbors(g, s, nb);
        x[rank[s] * 3 + 0] = (float)d;
        x[rank[s] * 3 + 1] = s < 2 ? 1.f : 0.f;
        x[rank[s] * 3 + 2] = g->maker_turn ? 1.f : 0.f;
        for (int i = 0; i < d && nb[i] < s; ++i) {
            edge_index[k] = rank[s];          edge_index[2 * E + k] = rank[nb[i]];
            edge_index[E + k] = rank[nb[i]];  edge_index[2 * E + E + k] = rank[s];
            ++k;
        }
    }
    *e2 = 2 * E;
    return n;
}

/* full adjacency dump for bit-exact comparison with the HIP builder: row-major nv x words */
void hexref_dump(const game_t* g, uint64_t* adj, uint8_t* alive) {
    memcpy(adj, g->adj, sizeof(uint64_t) * (size_t)g->nv * g->words);
    memcpy(alive, g->alive, (size_t)g->nv);
}
int hexref_words(const game_t* g) { return g->words; }
int hexref_nv(const game_t* g) { return g->nv; }
