// Batched Hex / Shannon node-switching environments on the GPU: one wavefront per env.
//
// Replaces the python game logic the RainbowDQN loop steps through (graph_game/shannon_node_switching_game.py:80-205,
// graph_game/hex_board_game.py:214-233 on graph-tool) and the graph->tensor conversion
// (GN0/util/convert_graph.py:60-130, torch_geometric Batch.from_data_list): reset / make_move /
// dead_and_captured / who_won / observe for num_envs lock-stepped games, emitting the batched observation
// (features, edge_index, backmap, ptr, and the sorted CSR the model kernels consume) directly in device memory.
//
// Data layout.  Per env: a symmetric adjacency BIT MATRIX adj[nv][W] (nv = size^2+2 vertices with their original
// ids, W = ceil(nv/64) 64-bit words), alive[nv], side to move, move count.  A step loads the env's matrix into LDS
// (Hex-11: 123 x 2 x 8 B = 2 KB), lane l owns vertices l, l+64, ...; set algebra is word-parallel, "is this set a
// clique" / "find the twin" are per-lane tests combined with wave ballots.  Integer/bit work: no MFMA, bound by LDS
// latency; results are bit-exact against oracle/env_ref.c (same canonical ascending order).
#include <vector>
#include "hexgnn_common.h"

namespace hexgnn {

constexpr int kMaxW = 10;   // up to 640 vertices (Hex-25 = 627)

struct EnvDev {
    int num_envs, size, nv, W, K;
    uint64_t* adj;        // [num_envs][nv][W]
    uint8_t* alive;       // [num_envs][nv]
    int* maker_turn;      // [num_envs]
    int* total_moves;     // [num_envs]
    short* resp_maker;    // [num_envs][nv]  (-1 = none)
    short* resp_breaker;  // [num_envs][nv]
    const uint64_t* start_adj;   // [nv][W]
};

struct Env {   // host handle
    EnvDev d;
    void* blob;
};

// WT = compile-time number of 64-bit words per vertex set.  WT < kMaxW instantiations are launched only for boards with
// exactly W == WT words (hexgnn_env_step's switch), so Wr() is the constant WT there (register-resident sets, row strides
// and word loops fully unrolled); WT == kMaxW is the generic fallback with the runtime W.
template <int WT> struct SetsT { uint64_t w[WT]; };

// ------------------------------------------------------------------------------------------------------------
// wave-level primitives on the LDS-resident game (blockDim.x == 64)
// ------------------------------------------------------------------------------------------------------------
template <int WT>
struct GameT {
    using Sets = SetsT<WT>;
    uint64_t* adj;     // LDS [nv][W]
    uint8_t* alive;    // LDS [nv]
    uint64_t* scr;     // LDS scratch [4][kMaxW]
    int nv, W, K, lane;
    __device__ __forceinline__ int Wr() const { if constexpr (WT == kMaxW) return W; else return WT; }
    bool maker_won;

    __device__ __forceinline__ uint64_t* row(int v) const { return adj + v * Wr(); }
    __device__ __forceinline__ void sync() const { __syncthreads(); }

    __device__ __forceinline__ Sets get_row(int v) const {
        Sets s;
#pragma unroll
        for (int w = 0; w < WT; ++w) s.w[w] = w < Wr() ? adj[v * Wr() + w] : 0ull;
        return s;
    }
    // word selection by compare chains: a dynamically indexed register array would be demoted to scratch memory
    __device__ __forceinline__ static uint64_t word(const Sets& s, int i) {
        uint64_t r = s.w[0];
#pragma unroll
        for (int w = 1; w < WT; ++w) r = (i == w) ? s.w[w] : r;
        return r;
    }
    __device__ __forceinline__ static bool has(const Sets& s, int v) { return (word(s, v >> 6) >> (v & 63)) & 1ull; }
    __device__ __forceinline__ static void clr(Sets& s, int v) {
        const uint64_t m = ~(1ull << (v & 63));
#pragma unroll
        for (int w = 0; w < WT; ++w) if ((v >> 6) == w) s.w[w] &= m;
    }
    __device__ __forceinline__ static void set(Sets& s, int v) {
        const uint64_t m = 1ull << (v & 63);
#pragma unroll
        for (int w = 0; w < WT; ++w) if ((v >> 6) == w) s.w[w] |= m;
    }
    __device__ __forceinline__ bool empty(const Sets& s) const {
        uint64_t o = 0;
#pragma unroll
        for (int w = 0; w < WT; ++w) o |= s.w[w];
        return o == 0;
    }
    // smallest element >= from, or -1 (uniform)
    __device__ __forceinline__ int next_bit(const Sets& s, int from) const {
        int found = -1;
#pragma unroll
        for (int w = WT - 1; w >= 0; --w) {       // descending, so the smallest matching word wins
            if (w < Wr() && w >= (from >> 6)) {
                uint64_t m = s.w[w];
                if (w == (from >> 6)) m &= ~0ull << (from & 63);
                if (m) found = w * 64 + __builtin_ctzll(m);
            }
        }
        return found;
    }
    // every vertex of s is adjacent to every other vertex of s (graph_game/utils.py:104-116)
    __device__ __forceinline__ bool is_clique(const Sets& s) const {
        bool ok = true;
        for (int k = 0; k < K; ++k) {
            const int x = lane + 64 * k;
            if (x < nv && has(s, x)) {
                const uint64_t* rx = row(x);
                for (int w = 0; w < Wr(); ++w) {
                    uint64_t need = s.w[w];
                    if (w == (x >> 6)) need &= ~(1ull << (x & 63));
                    if ((rx[w] & need) != need) ok = false;
                }
            }
        }
        return __all(ok);
    }
    // hide every edge of v and mark it removed (vp.f[v] = False)
    __device__ __forceinline__ void remove_vertex(int v) {
        sync();
        const uint64_t m = ~(1ull << (v & 63));
        for (int k = 0; k < K; ++k) {
            const int x = lane + 64 * k;
            if (x < nv) adj[x * Wr() + (v >> 6)] &= m;
        }
        sync();
        if (lane < Wr()) adj[v * Wr() + lane] = 0ull;
        if (lane == 0) alive[v] = 0;
        sync();
    }
    // OR of the per-lane contributions across the wave (butterfly over lanes: no LDS round trip, no barrier)
    __device__ __forceinline__ Sets wave_or(const Sets& mine, int /*slot*/) {
        Sets out;
#pragma unroll
        for (int w = 0; w < WT; ++w) {
            uint64_t v = w < Wr() ? mine.w[w] : 0ull;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) v |= __shfl_xor(v, off);
            out.w[w] = v;
        }
        return out;
    }

    // shannon_node_switching_game.py:80-116 without the final filtering of v.  Returns the change set of
    // _fix_teminal_connections (py:57-65).  Canonical ascending pair order (see oracle/env_ref.c): with a terminal
    // t among the neighbours every other neighbour is first wired to t, which makes every later pair "both touch t"
    // (skipped); with both terminals adjacent the maker has connected them and the game is over.
    __device__ __forceinline__ Sets maker_connect(int v) {
        Sets change;
#pragma unroll
        for (int w = 0; w < WT; ++w) change.w[w] = 0ull;
        const Sets nb = get_row(v);
        const bool t0 = has(nb, 0), t1 = has(nb, 1);
        if (t0 && t1) {
            sync();
            if (lane == 0) { adj[0 * Wr()] |= 2ull; adj[1 * Wr()] |= 1ull; }
            maker_won = true;
            sync();
            return change;
        }
        if (!t0 && !t1) {
            const Sets n0 = get_row(0), n1 = get_row(1);
            sync();
            for (int k = 0; k < K; ++k) {
                const int x = lane + 64 * k;
                if (x < nv && has(nb, x)) {
                    const bool a0 = has(n0, x), a1 = has(n1, x);
                    for (int w = 0; w < Wr(); ++w) {
                        uint64_t m = nb.w[w];
                        if (w == (x >> 6)) m &= ~(1ull << (x & 63));
                        if (a0) m &= ~n0.w[w];
                        if (a1) m &= ~n1.w[w];
                        adj[x * Wr() + w] |= m;
                    }
                }
            }
            sync();
            return change;
        }
        const int t = t0 ? 0 : 1;
        sync();
        for (int k = 0; k < K; ++k) {
            const int x = lane + 64 * k;
            if (x < nv && x != t && has(nb, x)) adj[x * Wr()] |= 1ull << t;
        }
        if (lane < Wr()) {
            uint64_t m = word(nb, lane);
            if (lane == 0) m &= ~(1ull << t);
            adj[t * Wr() + lane] |= m;
        }
        sync();
        // _fix_teminal_connections(t): drop every edge between two neighbours of t
        const Sets nt = get_row(t);
        Sets mine;
#pragma unroll
        for (int w = 0; w < WT; ++w) mine.w[w] = 0ull;
        for (int k = 0; k < K; ++k) {
            const int x = lane + 64 * k;
            if (x < nv && has(nt, x)) {
                uint64_t any = 0;
                for (int w = 0; w < Wr(); ++w) {
                    const uint64_t rem = adj[x * Wr() + w] & nt.w[w];
                    any |= rem;
                    adj[x * Wr() + w] &= ~nt.w[w];
                }
                if (any) set(mine, x);
            }
        }
        return wave_or(mine, 0);
    }

    // ---- single-lane versions of the tests (every lane screens a different candidate vertex) ----------------------
    __device__ __forceinline__ bool lane_is_clique(const Sets& s) const {
        bool ok = true;
#pragma unroll
        for (int w = 0; w < WT; ++w) {
            if (w < Wr()) {
                uint64_t bits = s.w[w];
                while (bits && ok) {
                    const int bpos = __builtin_ctzll(bits);
                    bits &= bits - 1;
                    const uint64_t* ry = row(64 * w + bpos);
#pragma unroll
                    for (int u = 0; u < WT; ++u) {
                        if (u < Wr()) {
                            uint64_t need = s.w[u];
                            if (u == w) need &= ~(1ull << bpos);
                            if ((ry[u] & need) != need) ok = false;
                        }
                    }
                }
            }
        }
        return ok;
    }
    // What the sequential pass would do with vertex x in the CURRENT graph: 0 nothing, 1 dead, 2 maker capture (partner =
    // twin), 3 breaker capture (partner = neighbour).  Same tests, same priorities, same ascending candidate order.
    __device__ __forceinline__ int lane_screen(int x, int& partner) const {
        const Sets ns = get_row(x);
        if (lane_is_clique(ns)) return 1;
        const int one = next_bit(ns, 0);
        Sets cs = get_row(one);
        set(cs, one);
        int first = -1;
        bool hit_one = false;
#pragma unroll
        for (int w = 0; w < WT; ++w) {
            if (w < Wr()) {
                uint64_t bits = cs.w[w];
                while (bits) {
                    const int bpos = __builtin_ctzll(bits);
                    bits &= bits - 1;
                    const int c = 64 * w + bpos;
                    if (c < 2 || c == x || !alive[c]) continue;
                    const uint64_t* rc = row(c);
                    bool hit = true;
#pragma unroll
                    for (int u = 0; u < WT; ++u) {
                        if (u < Wr()) {
                            uint64_t ra = rc[u], rb = ns.w[u];
                            if (u == (x >> 6)) ra &= ~(1ull << (x & 63));
                            if (u == (c >> 6)) rb &= ~(1ull << (c & 63));
                            if (ra != rb) hit = false;
                        }
                    }
                    if (hit) { if (c == one) hit_one = true; if (first < 0) first = c; }
                }
            }
        }
        const int twin = hit_one ? one : first;
        if (twin >= 0) { partner = twin; return 2; }
        for (int nbr = next_bit(ns, 2); nbr >= 0; nbr = next_bit(ns, nbr + 1)) {
            Sets wm = get_row(nbr), wh = ns;
            clr(wm, x);
            clr(wh, nbr);
            if (lane_is_clique(wm) && lane_is_clique(wh)) { partner = nbr; return 3; }
        }
        return 0;
    }

    // shannon_node_switching_game.py:119-196 with iterate=True (canonical order: oracle/env_ref.c).  The reference visits
    // the candidates one by one; most of them need no action.  Here every lane screens one candidate against the current
    // graph, the smallest vertex that needs an action is handled by the whole wave, and the candidates above it are screened
    // again on the changed graph -- exactly the sequence of actions of the ascending sequential pass, in (#actions + 1)
    // screening rounds instead of one wave-wide evaluation per candidate.
    __device__ __forceinline__ void dead_and_captured(Sets consider, short* resp_m, short* resp_b) {
        while (!empty(consider)) {
            Sets big;
#pragma unroll
            for (int w = 0; w < WT; ++w) big.w[w] = 0ull;
            int from = 2;
            while (true) {
                sync();
                int code = 0, partner = -1, node = -1;
                for (int k = 0; k < K && code == 0; ++k) {
                    const int x = lane + 64 * k;
                    int c = 0, p = -1;
                    if (x < nv && x >= from && has(consider, x) && alive[x]) c = lane_screen(x, p);
                    const uint64_t bal = __ballot(c != 0);
                    if (bal) {
                        const int srcl = __builtin_ctzll(bal);
                        node = 64 * k + srcl;
                        code = __shfl(c, srcl);
                        partner = __shfl(p, srcl);
                    }
                }
                if (code == 0) break;
                from = node + 1;
                const Sets ns = get_row(node);
                if (code == 1) {                              // dead
#pragma unroll
                    for (int w = 0; w < WT; ++w) big.w[w] |= ns.w[w];
                    remove_vertex(node);
                } else if (code == 2) {                       // maker capture: twin with the same neighbourhood
                    const int twin = partner;
                    Sets wm = get_row(twin);
                    clr(wm, node);
                    if (lane == 0) { resp_m[node] = (short)twin; resp_m[twin] = (short)node; }
                    remove_vertex(twin);
                    const Sets ch = maker_connect(node);
                    remove_vertex(node);
#pragma unroll
                    for (int w = 0; w < WT; ++w) big.w[w] |= wm.w[w] | ch.w[w];
                } else {                                      // breaker capture
                    const int nbr = partner;
                    Sets wm = get_row(nbr), wh = ns;
                    clr(wm, node);
                    clr(wh, nbr);
#pragma unroll
                    for (int w = 0; w < WT; ++w) big.w[w] |= wm.w[w] | wh.w[w];
                    if (lane == 0) { resp_b[node] = (short)nbr; resp_b[nbr] = (short)node; }
                    remove_vertex(node);
                    remove_vertex(nbr);
                }
            }
            consider = big;
        }
    }

    // 0 = maker, 1 = breaker, -1 = undecided (py:199-205): edge t0-t1, else reachability t0 -> t1
    __device__ __forceinline__ int who_won() {
        sync();
        if (maker_won || (adj[0] & 2ull)) return 0;
        Sets reach, frontier;
#pragma unroll
        for (int w = 0; w < WT; ++w) reach.w[w] = 0ull;
        reach.w[0] = 1ull;
        frontier = reach;
        for (int it = 0; it < nv; ++it) {
            Sets mine;
#pragma unroll
            for (int w = 0; w < WT; ++w) mine.w[w] = 0ull;
            for (int k = 0; k < K; ++k) {
                const int x = lane + 64 * k;
                if (x < nv && has(frontier, x))
                    for (int w = 0; w < Wr(); ++w) mine.w[w] |= adj[x * Wr() + w];
            }
            const Sets nr = wave_or(mine, 1);
            bool grew = false;
#pragma unroll
            for (int w = 0; w < WT; ++w) {
                frontier.w[w] = nr.w[w] & ~reach.w[w];        // newly reached vertices only
                grew |= frontier.w[w] != 0ull;
                reach.w[w] |= nr.w[w];
            }
            if (has(reach, 1)) return -1;
            if (!grew) break;
        }
        return 1;
    }
};
using Game = GameT<kMaxW>;

// observation source: an array of board states (the live envs, or a replay ring) + which of them to emit
struct StateSrc {
    int nv, W, K;
    const uint64_t* adj;      // [num_states][nv][W]
    const uint8_t* alive;     // [num_states][nv]
    const int* maker_turn;    // [num_states] int32 side flags, or null
    const uint8_t* side_u8;   // [num_states] u8 side flags (used when maker_turn is null)
    const int* index;         // [k] state picked for output graph i, or null (identity)
};

template <class G>
__device__ __forceinline__ void load_game(G& g, const EnvDev& d, int env, char* lds) {
    g.nv = d.nv; g.W = d.W; g.K = d.K; g.lane = threadIdx.x; g.maker_won = false;
    g.adj = reinterpret_cast<uint64_t*>(lds);
    g.scr = g.adj + (size_t)d.nv * d.W;
    g.alive = reinterpret_cast<uint8_t*>(g.scr + 4 * kMaxW);
    const uint64_t* src = d.adj + (size_t)env * d.nv * d.W;
    for (int i = threadIdx.x; i < d.nv * d.W; i += 64) g.adj[i] = src[i];
    for (int i = threadIdx.x; i < d.nv; i += 64) g.alive[i] = d.alive[(size_t)env * d.nv + i];
    __syncthreads();
}
template <class G>
__device__ __forceinline__ void store_game(const G& g, const EnvDev& d, int env) {
    __syncthreads();
    uint64_t* dst = d.adj + (size_t)env * d.nv * d.W;
    for (int i = threadIdx.x; i < d.nv * d.W; i += 64) dst[i] = g.adj[i];
    for (int i = threadIdx.x; i < d.nv; i += 64) d.alive[(size_t)env * d.nv + i] = g.alive[i];
}
template <class G>
__device__ __forceinline__ void reset_game_lds(G& g, const EnvDev& d) {
    __syncthreads();
    for (int i = threadIdx.x; i < d.nv * d.W; i += 64) g.adj[i] = d.start_adj[i];
    for (int i = threadIdx.x; i < d.nv; i += 64) g.alive[i] = 1;
    g.maker_won = false;
    __syncthreads();
}
// alive count and directed edge count (uniform)
template <class G>
__device__ __forceinline__ void count_game(const G& g, int* n_alive, int* n_dir_edges) {
    int a = 0, e = 0;
    for (int k = 0; k < g.K; ++k) {
        const int x = g.lane + 64 * k;
        if (x < g.nv && g.alive[x]) {
            ++a;
            for (int w = 0; w < g.W; ++w) e += __popcll(g.adj[x * g.W + w]);
        }
    }
    for (int off = 32; off >= 1; off >>= 1) { a += __shfl_xor(a, off); e += __shfl_xor(e, off); }
    *n_alive = a; *n_dir_edges = e;
}

// ------------------------------------------------------------------------------------------------------------
// kernels: one 64-thread workgroup per env
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void env_reset_kernel(EnvDev d, const uint8_t* __restrict__ mask, int maker_turn,
                                                     int* __restrict__ sizes /*[num_envs][2]*/) {
    const int env = blockIdx.x;
    const bool doit = !mask || mask[env];
    if (doit) {
        uint64_t* dst = d.adj + (size_t)env * d.nv * d.W;
        for (int i = threadIdx.x; i < d.nv * d.W; i += 64) dst[i] = d.start_adj[i];
        for (int i = threadIdx.x; i < d.nv; i += 64) {
            d.alive[(size_t)env * d.nv + i] = 1;
            d.resp_maker[(size_t)env * d.nv + i] = -1;
            d.resp_breaker[(size_t)env * d.nv + i] = -1;
        }
        if (threadIdx.x == 0) { d.maker_turn[env] = maker_turn; d.total_moves[env] = 0; }
    }
    if (sizes && doit && threadIdx.x == 0) {
        int e = 0;
        for (int i = 0; i < d.nv * d.W; ++i) e += __popcll(d.start_adj[i]);
        sizes[2 * env] = d.nv; sizes[2 * env + 1] = e;
    }
}

// result record per env: [winner(-1/0/1), length, n_alive, n_directed_edges, error]
template <int WT>
__global__ __launch_bounds__(64) void env_step_kernel(EnvDev d, const int* __restrict__ actions, int remove_dc,
                                                    int auto_reset, int reset_maker_turn, int* __restrict__ result) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int env = blockIdx.x;
    GameT<WT> g;
    using Sets = typename GameT<WT>::Sets;
    load_game(g, d, env, lds);
    short* rm = d.resp_maker + (size_t)env * d.nv;
    short* rb = d.resp_breaker + (size_t)env * d.nv;
    int* res = result + 5 * env;
    const int v = actions[env];
    int err = 0;
    if (v < 2 || v >= d.nv || !g.alive[v]) err = 1;
    int mt = d.maker_turn[env], moves = d.total_moves[env];
    int winner = -1;
    if (!err) {
        ++moves;
        Sets consider;
#pragma unroll
        for (int w = 0; w < WT; ++w) consider.w[w] = 0ull;
        if (mt) consider = g.maker_connect(v);
        const Sets nbv = g.get_row(v);
#pragma unroll
        for (int w = 0; w < WT; ++w) consider.w[w] |= nbv.w[w];
        g.remove_vertex(v);
        mt = !mt;
        if (remove_dc && !g.maker_won) g.dead_and_captured(consider, rm, rb);
        winner = g.who_won();
    }
    int length = moves;
    if (winner >= 0 && auto_reset) {
        reset_game_lds(g, d);
        for (int i = threadIdx.x; i < d.nv; i += 64) { rm[i] = -1; rb[i] = -1; }
        mt = reset_maker_turn;
        moves = 0;
    }
    int na, ne;
    count_game(g, &na, &ne);
    store_game(g, d, env);
    if (threadIdx.x == 0) {
        d.maker_turn[env] = mt; d.total_moves[env] = moves;
        res[0] = winner; res[1] = length; res[2] = na; res[3] = ne; res[4] = err;
    }
}

// Observation of every env into batched buffers (convert_graph.py:77-122, old_style=True; Batch.from_data_list):
//   x [N][3] = (degree, is_terminal, maker_to_move);  backmap [N] rank -> vertex id (int64)
//   edge_local [2][E]: per graph, first the E_g/2 edges (s > t, sorted by (s,t)) then the flipped copies, LOCAL ranks
//   edge_global [2][E]: the same with the graph's node offset added (Batch.edge_index)
//   rowptr [N+1], col [E]: sorted CSR over global node ids (== what hexgnn_csr_build would produce), invdeg [N]
// node_off / edge_off: exclusive prefix sums of the per-env sizes (computed by the caller from the step result).
__global__ __launch_bounds__(64) void env_observe_kernel(StateSrc d, const int* __restrict__ node_off,
                                                       const int* __restrict__ edge_off, float* __restrict__ x,
                                                       int64_t* __restrict__ backmap, int64_t* __restrict__ edge_local,
                                                       int64_t* __restrict__ edge_global, int64_t e_total,
                                                       int* __restrict__ rowptr, int* __restrict__ col,
                                                       float* __restrict__ invdeg, int64_t* __restrict__ batch_vec,
                                                       int write_rowptr_end) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int env = blockIdx.x;                               // output graph
    const int src = d.index ? d.index[env] : env;             // state it is built from
    Game g;
    g.nv = d.nv; g.W = d.W; g.K = d.K; g.lane = threadIdx.x; g.maker_won = false;
    g.adj = reinterpret_cast<uint64_t*>(lds);
    g.scr = g.adj + (size_t)d.nv * d.W;
    g.alive = reinterpret_cast<uint8_t*>(g.scr + 4 * kMaxW);
    {
        const uint64_t* sa = d.adj + (size_t)src * d.nv * d.W;
        for (int i = threadIdx.x; i < d.nv * d.W; i += 64) g.adj[i] = sa[i];
        for (int i = threadIdx.x; i < d.nv; i += 64) g.alive[i] = d.alive[(size_t)src * d.nv + i];
        __syncthreads();
    }
    short* rank = reinterpret_cast<short*>(g.alive + ((d.nv + 15) / 16) * 16);   // [nv]
    int* lowoff = reinterpret_cast<int*>(rank + ((d.nv + 7) / 8) * 8);           // [nv]: offset of the (s>t) edges of s
    int* rowoff = lowoff + d.nv;                                                    // [nv]: CSR row start (local)
    const int lane = threadIdx.x;
    const int n0 = node_off[env], e0 = edge_off[env];
    const int ne = edge_off[env + 1] - e0;       // directed edges of this graph
    const int half = ne / 2;
    const float side = (d.maker_turn ? d.maker_turn[src] != 0 : d.side_u8[src] != 0) ? 1.f : 0.f;
    // ranks + per-vertex counts via running prefix over 64-vertex groups
    int run_rank = 0, run_low = 0, run_row = 0;
    for (int k = 0; k < g.K; ++k) {
        const int v = lane + 64 * k;
        const bool al = v < d.nv && g.alive[v];
        int deg = 0, low = 0;
        if (al) {
            for (int w = 0; w < g.W; ++w) {
                const uint64_t r = g.adj[v * g.W + w];
                deg += __popcll(r);
                uint64_t lm = w < (v >> 6) ? ~0ull : (w == (v >> 6) ? ((1ull << (v & 63)) - 1ull) : 0ull);
                low += __popcll(r & lm);
            }
        }
        const uint64_t bal = __ballot(al);
        const int myrank = run_rank + __popcll(bal & ((1ull << lane) - 1ull));
        // exclusive prefix sums of low / deg across the wave
        int plow = low, pdeg = deg;
        for (int off = 1; off < 64; off <<= 1) {
            const int a = __shfl_up(plow, off), b = __shfl_up(pdeg, off);
            if (lane >= off) { plow += a; pdeg += b; }
        }
        if (al) {
            rank[v] = (short)myrank;
            lowoff[v] = run_low + plow - low;
            rowoff[v] = run_row + pdeg - deg;
            const int gn = n0 + myrank;
            x[(size_t)gn * 3 + 0] = (float)deg;
            x[(size_t)gn * 3 + 1] = v < 2 ? 1.f : 0.f;
            x[(size_t)gn * 3 + 2] = side;
            backmap[gn] = v;
            batch_vec[gn] = env;
            rowptr[gn] = e0 + rowoff[v];
            invdeg[gn] = 1.f / (float)max(deg, 1);
        } else if (v < d.nv) rank[v] = -1;
        run_rank += __popcll(bal);
        run_low += __shfl(plow, 63);
        run_row += __shfl(pdeg, 63);
    }
    __syncthreads();
    if (write_rowptr_end && env == (int)gridDim.x - 1 && lane == 0) rowptr[n0 + run_rank] = e0 + run_row;
    // edges
    for (int k = 0; k < g.K; ++k) {
        const int v = lane + 64 * k;
        if (v < d.nv && g.alive[v]) {
            const int rs = rank[v];
            int lo = lowoff[v], ro = rowoff[v];
            for (int w = 0; w < g.W; ++w) {
                uint64_t r = g.adj[v * g.W + w];
                while (r) {
                    const int u = w * 64 + __builtin_ctzll(r);
                    r &= r - 1;
                    const int ru = rank[u];
                    col[e0 + ro] = n0 + ru;
                    ++ro;
                    if (u < v) {
                        const int64_t p = e0 + lo;
                        edge_local[p] = rs;               edge_local[e_total + p] = ru;
                        edge_local[p + half] = ru;        edge_local[e_total + p + half] = rs;
                        edge_global[p] = n0 + rs;         edge_global[e_total + p] = n0 + ru;
                        edge_global[p + half] = n0 + ru;  edge_global[e_total + p + half] = n0 + rs;
                        ++lo;
                    }
                }
            }
        }
    }
}

// small state readers (tests / host mirrors)
__global__ void env_export_kernel(EnvDev d, uint64_t* __restrict__ adj, uint8_t* __restrict__ alive,
                                  int* __restrict__ maker_turn, int* __restrict__ total_moves,
                                  short* __restrict__ resp_maker, short* __restrict__ resp_breaker) {
    const size_t tot = (size_t)d.num_envs * d.nv;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < tot * d.W; i += (size_t)gridDim.x * blockDim.x) adj[i] = d.adj[i];
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < tot; i += (size_t)gridDim.x * blockDim.x) {
        alive[i] = d.alive[i];
        if (resp_maker) resp_maker[i] = d.resp_maker[i];
        if (resp_breaker) resp_breaker[i] = d.resp_breaker[i];
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < d.num_envs; i += gridDim.x * blockDim.x) {
        maker_turn[i] = d.maker_turn[i];
        total_moves[i] = d.total_moves[i];
    }
}

// inverse of env_export_kernel: overwrite the whole env state (checkpoint restore, rollback after a captured warm-up)
__global__ void env_import_kernel(EnvDev d, const uint64_t* __restrict__ adj, const uint8_t* __restrict__ alive,
                                  const int* __restrict__ maker_turn, const int* __restrict__ total_moves,
                                  const short* __restrict__ resp_maker, const short* __restrict__ resp_breaker) {
    const size_t tot = (size_t)d.num_envs * d.nv;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < tot * d.W; i += (size_t)gridDim.x * blockDim.x) d.adj[i] = adj[i];
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < tot; i += (size_t)gridDim.x * blockDim.x) {
        d.alive[i] = alive[i];
        d.resp_maker[i] = resp_maker[i];
        d.resp_breaker[i] = resp_breaker[i];
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < d.num_envs; i += gridDim.x * blockDim.x) {
        d.maker_turn[i] = maker_turn[i];
        d.total_moves[i] = total_moves[i];
    }
}

__global__ void env_set_turn_kernel(EnvDev d, int maker_turn) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < d.num_envs) d.maker_turn[i] = maker_turn;
}

static size_t env_lds_bytes(const EnvDev& d) {
    size_t b = sizeof(uint64_t) * ((size_t)d.nv * d.W + 4 * kMaxW);
    b += (size_t)((d.nv + 15) / 16) * 16;                 // alive
    b += sizeof(short) * (size_t)((d.nv + 7) / 8) * 8;    // rank
    b += sizeof(int) * 2 * (size_t)d.nv;                  // lowoff, rowoff
    return align_up(b, 16);
}

template <int WT>
static void launch_env_step(const EnvDev& d, const int* actions, int remove_dc, int auto_reset, int reset_maker_turn,
                            int* result, hipStream_t st) {
    static bool once = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&env_step_kernel<WT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        return true;
    }();
    (void)once;
    env_step_kernel<WT><<<d.num_envs, 64, env_lds_bytes(d), st>>>(d, actions, remove_dc, auto_reset, reset_maker_turn, result);
}


}  // namespace hexgnn

using namespace hexgnn;

// exclusive prefix sums of the per-env graph sizes a step reported (result[env] = {winner, moves, nodes, edges, illegal}):
// the batched observation's node / edge offsets stay on the device, so a rollout never reads sizes back
__global__ __launch_bounds__(64) void env_offsets_kernel(int k, const int* __restrict__ result, int* __restrict__ node_off,
                                                         int* __restrict__ edge_off) {
    const int lane = threadIdx.x;
    int run_n = 0, run_e = 0;
    for (int base = 0; base < k; base += 64) {
        const int i = base + lane;
        const int vn = i < k ? result[i * 5 + 2] : 0, ve = i < k ? result[i * 5 + 3] : 0;
        int pn = vn, pe = ve;
        for (int off = 1; off < 64; off <<= 1) {
            const int a = __shfl_up(pn, off), b = __shfl_up(pe, off);
            if (lane >= off) { pn += a; pe += b; }
        }
        if (i < k) { node_off[i] = run_n + pn - vn; edge_off[i] = run_e + pe - ve; }
        run_n += __shfl(pn, 63);
        run_e += __shfl(pe, 63);
    }
    if (lane == 0) { node_off[k] = run_n; edge_off[k] = run_e; }
}

extern "C" {

int hexgnn_env_create(int num_envs, int hex_size, hexgnn_env** out) {
    if (!out || num_envs < 1 || hex_size < 2) return HEXGNN_EINVAL;
    const int nv = hex_size * hex_size + 2, W = (nv + 63) / 64;
    if (W > kMaxW) return HEXGNN_EUNSUPPORTED;
    Env* e = new Env();
    EnvDev& d = e->d;
    d.num_envs = num_envs; d.size = hex_size; d.nv = nv; d.W = W; d.K = (nv + 63) / 64;
    // start graph (graph_game/hex_board_game.py:214-233, redgraph=True, no_worthless_edges=True)
    std::vector<uint64_t> start((size_t)nv * W, 0ull);
    auto add = [&](int a, int b) {
        start[(size_t)a * W + (b >> 6)] |= 1ull << (b & 63);
        start[(size_t)b * W + (a >> 6)] |= 1ull << (a & 63);
    };
    const int n = hex_size, sq = n * n;
    for (int i = 0; i < sq; ++i) {
        const int v = i + 2;
        if (i < n) add(v, 0);
        if (i / n == n - 1) add(v, 1);
        if (i % n > 0 && n <= i && i <= sq - n) add(v, v - 1);
        if (i >= n) {
            add(v, v - n);
            if (i % n != n - 1) add(v, v - n + 1);
        }
    }
    const size_t tot = (size_t)num_envs * nv;
    size_t off = 0;
    const size_t o_adj = off; off += align_up(sizeof(uint64_t) * tot * W, 256);
    const size_t o_start = off; off += align_up(sizeof(uint64_t) * (size_t)nv * W, 256);
    const size_t o_alive = off; off += align_up(tot, 256);
    const size_t o_mt = off; off += align_up(sizeof(int) * num_envs, 256);
    const size_t o_tm = off; off += align_up(sizeof(int) * num_envs, 256);
    const size_t o_rm = off; off += align_up(sizeof(short) * tot, 256);
    const size_t o_rb = off; off += align_up(sizeof(short) * tot, 256);
    char* blob = nullptr;
    if (hipMalloc(&blob, off) != hipSuccess) { delete e; return HEXGNN_EHIP; }
    e->blob = blob;
    d.adj = (uint64_t*)(blob + o_adj); d.start_adj = (const uint64_t*)(blob + o_start);
    d.alive = (uint8_t*)(blob + o_alive); d.maker_turn = (int*)(blob + o_mt); d.total_moves = (int*)(blob + o_tm);
    d.resp_maker = (short*)(blob + o_rm); d.resp_breaker = (short*)(blob + o_rb);
    if (hipMemcpy(blob + o_start, start.data(), sizeof(uint64_t) * (size_t)nv * W, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(blob); delete e; return HEXGNN_EHIP;
    }
    env_reset_kernel<<<num_envs, 64, 0, 0>>>(d, nullptr, 1, nullptr);
    if (hipDeviceSynchronize() != hipSuccess) { (void)hipFree(blob); delete e; return HEXGNN_EHIP; }
    *out = reinterpret_cast<hexgnn_env*>(e);
    return HEXGNN_OK;
}

void hexgnn_env_destroy(hexgnn_env* h) {
    if (!h) return;
    Env* e = reinterpret_cast<Env*>(h);
    (void)hipFree(e->blob);
    delete e;
}

int hexgnn_env_num_vertices(const hexgnn_env* h) { return h ? reinterpret_cast<const Env*>(h)->d.nv : -1; }
int hexgnn_env_words(const hexgnn_env* h) { return h ? reinterpret_cast<const Env*>(h)->d.W : -1; }

int hexgnn_env_reset(hexgnn_env* h, const uint8_t* mask, int maker_turn, int* sizes, hexgnn_stream_t stream_) {
    if (!h) return HEXGNN_EINVAL;
    Env* e = reinterpret_cast<Env*>(h);
    env_reset_kernel<<<e->d.num_envs, 64, 0, (hipStream_t)stream_>>>(e->d, mask, maker_turn ? 1 : 0, sizes);
    return check_launch();
}

int hexgnn_env_set_maker_turn(hexgnn_env* h, int maker_turn, hexgnn_stream_t stream_) {
    if (!h) return HEXGNN_EINVAL;
    Env* e = reinterpret_cast<Env*>(h);
    env_set_turn_kernel<<<(e->d.num_envs + 255) / 256, 256, 0, (hipStream_t)stream_>>>(e->d, maker_turn ? 1 : 0);
    return check_launch();
}

int hexgnn_env_step(hexgnn_env* h, const int* actions, int remove_dead_and_captured, int auto_reset,
                    int reset_maker_turn, int* result, hexgnn_stream_t stream_) {
    if (!h || !actions || !result) return HEXGNN_EINVAL;
    Env* e = reinterpret_cast<Env*>(h);
    hipStream_t st = (hipStream_t)stream_;
    const int rm = reset_maker_turn ? 1 : 0;
    // vertex sets as compile-time register arrays for the common board sizes (W = ceil((n*n+2)/64): Hex <= 7: 1,
    // Hex 8-11: 2, Hex 12-13: 3, Hex 14-15: 4); larger boards take the generic runtime-W instantiation
    switch (e->d.W) {
        case 1: launch_env_step<1>(e->d, actions, remove_dead_and_captured, auto_reset, rm, result, st); break;
        case 2: launch_env_step<2>(e->d, actions, remove_dead_and_captured, auto_reset, rm, result, st); break;
        case 3: launch_env_step<3>(e->d, actions, remove_dead_and_captured, auto_reset, rm, result, st); break;
        case 4: launch_env_step<4>(e->d, actions, remove_dead_and_captured, auto_reset, rm, result, st); break;
        default: launch_env_step<kMaxW>(e->d, actions, remove_dead_and_captured, auto_reset, rm, result, st); break;
    }
    return check_launch();
}

int hexgnn_env_observe(hexgnn_env* h, const int* node_off, const int* edge_off, int64_t e_total, float* x,
                       int64_t* backmap, int64_t* edge_local, int64_t* edge_global, int* rowptr, int* col,
                       float* invdeg, int64_t* batch_vec, hexgnn_stream_t stream_) {
    if (!h || !node_off || !edge_off || !x || !backmap || !edge_local || !edge_global || !rowptr || !col ||
        !invdeg || !batch_vec || e_total < 0)
        return HEXGNN_EINVAL;
    Env* e = reinterpret_cast<Env*>(h);
    static bool once = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&env_observe_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        return true;
    }();
    (void)once;
    StateSrc src;
    src.nv = e->d.nv; src.W = e->d.W; src.K = e->d.K;
    src.adj = e->d.adj; src.alive = e->d.alive; src.maker_turn = e->d.maker_turn; src.side_u8 = nullptr; src.index = nullptr;
    env_observe_kernel<<<e->d.num_envs, 64, env_lds_bytes(e->d), (hipStream_t)stream_>>>(
        src, node_off, edge_off, x, backmap, edge_local, edge_global, e_total, rowptr, col, invdeg, batch_vec, 1);
    return check_launch();
}

int hexgnn_states_observe(int hex_size, int k, const uint64_t* adj, const uint8_t* alive, const uint8_t* side,
                          const int* index, const int* node_off, const int* edge_off, int64_t e_total, float* x,
                          int64_t* backmap, int64_t* edge_local, int64_t* edge_global, int* rowptr, int* col,
                          float* invdeg, int64_t* batch_vec, hexgnn_stream_t stream_) {
    if (hex_size < 2 || k < 0 || !adj || !alive || !side || !node_off || !edge_off || !x || !backmap || !edge_local ||
        !edge_global || !rowptr || !col || !invdeg || !batch_vec || e_total < 0)
        return HEXGNN_EINVAL;
    const int nv = hex_size * hex_size + 2, W = (nv + 63) / 64;
    if (W > kMaxW) return HEXGNN_EUNSUPPORTED;
    if (k == 0) return HEXGNN_OK;
    static bool once = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&env_observe_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        return true;
    }();
    (void)once;
    EnvDev tmp;
    tmp.nv = nv; tmp.W = W; tmp.K = (nv + 63) / 64;
    StateSrc src;
    src.nv = nv; src.W = W; src.K = tmp.K;
    src.adj = adj; src.alive = alive; src.maker_turn = nullptr; src.side_u8 = side; src.index = index;
    env_observe_kernel<<<k, 64, env_lds_bytes(tmp), (hipStream_t)stream_>>>(
        src, node_off, edge_off, x, backmap, edge_local, edge_global, e_total, rowptr, col, invdeg, batch_vec, 1);
    return check_launch();
}

int hexgnn_env_import(hexgnn_env* h, const uint64_t* adj, const uint8_t* alive, const int* maker_turn,
                      const int* total_moves, const int16_t* resp_maker, const int16_t* resp_breaker, hexgnn_stream_t stream_) {
    if (!h || !adj || !alive || !maker_turn || !total_moves || !resp_maker || !resp_breaker) return HEXGNN_EINVAL;
    Env* e = reinterpret_cast<Env*>(h);
    env_import_kernel<<<64, 256, 0, (hipStream_t)stream_>>>(e->d, adj, alive, maker_turn, total_moves,
                                                          (const short*)resp_maker, (const short*)resp_breaker);
    return check_launch();
}

int hexgnn_env_offsets(int k, const int* result, int* node_off, int* edge_off, hexgnn_stream_t stream_) {
    if (k < 0 || !node_off || !edge_off || (k > 0 && !result)) return HEXGNN_EINVAL;
    env_offsets_kernel<<<1, 64, 0, (hipStream_t)stream_>>>(k, result, node_off, edge_off);
    return check_launch();
}

int hexgnn_env_export(hexgnn_env* h, uint64_t* adj, uint8_t* alive, int* maker_turn, int* total_moves,
                      int16_t* resp_maker, int16_t* resp_breaker, hexgnn_stream_t stream_) {
    if (!h || !adj || !alive || !maker_turn || !total_moves) return HEXGNN_EINVAL;
    Env* e = reinterpret_cast<Env*>(h);
    env_export_kernel<<<64, 256, 0, (hipStream_t)stream_>>>(e->d, adj, alive, maker_turn, total_moves,
                                                          (short*)resp_maker, (short*)resp_breaker);
    return check_launch();
}

}  // extern "C"
