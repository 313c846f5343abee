// Whole modern_two_headed Q-network per launch: one 512-thread workgroup per board graph (<= 128 nodes).
//
// Reference path: DuellingTwoHeaded.forward (GN0/models.py:537-584) = CachifiedGNN body (261-294) -> head gnn ->
// HeadNetwork tail (374-384) -> dueling combine (571-584), and its autograd backward.
//
// MI355X design.  A Hex-11 graph is 123 x 110 fp32 = 54 KB: the node features of ALL layers stay in the CU's
// 160 KB LDS, so neighbour gathers are LDS reads and nothing but the saved activations goes to HBM.
//   LDS:  [ W half A | W half B | node rows 129 x (HP+4) (row 128 all zero) | CSR (u16 rowptr, u8 col) | maxima | bias row ]
//   wave w owns rows 16w..16w+15; lane (r = l&15, g = l>>4) holds, for its row r, the 4-float feature chunks
//   {16c+4g..+3}, c < NT.  MFMAs run with SWAPPED operands (a = packed weights, b = row fragment): the tile comes
//   out transposed, i.e. in the SAME lane layout, so a layer's output registers are the next layer's self operand.
//   Per layer the [agg|x] contraction is split in two K phases, self half (W_r) first, then the aggregate half (W_l);
//   the layer's other work (LDS gather, saved-tensor stores / loads, LDS-DMA of the weight half the other phase needs)
//   is issued in small pieces between the MFMA groups of the two contractions (two barriers per layer).
//   fp32 in / fp32 accumulate (v_mfma_f32_16x16x4_f32): exact fmaf chains, deterministic.
#pragma once
#include <type_traits>
#include <utility>
#include "hexgnn_internal.h"
#include "hexgnn_memops.h"

namespace hexgnn {

constexpr int kMaxL = kMaxLayers;
constexpr int kRows = 128;            // rows per workgroup
constexpr int kLdsBytes = 160 * 1024;

// Profiling aid (build with `make STAMPS=1`, never shipped): lane 0 of every wave of workgroup 0 records s_memtime at the
// marked points of each layer; tools/stamps.py prints the per-wave timeline DESIGN.md quotes.
#ifdef HEXGNN_STAMPS
constexpr int kStampPoints = 10;
static __device__ unsigned long long g_qstamps[2][kMaxLayers + 2][kStampPoints][8];
#define QSTAMP(k, l, p) do { if (blockIdx.x == 0 && lane == 0) g_qstamps[k][l][p][wave] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define QSTAMP(k, l, p) do {} while (0)
#endif

struct QFwdArgs {
    int n, b, c_in, H, L, mode, x_stride, need_backward;
    const int* gptr; const int* rowptr; const int* col; const float* invdeg;
    const float* x;
    const char* wpack; size_t fwd_off[kMaxL]; size_t bias_off[kMaxL];
    float* acts; char* saved; size_t agg_off[kMaxL];
    const float* lin_w; const float* lin_b; const float* v0_w; const float* v0_b; const float* v1_w; const float* v1_b;
    float* adv_raw; float* pooled; int* amax; int* amin; float* z; float* vraw;
    float* q; float* out_v; int* status;
    unsigned* xmax;   // [L] bit patterns of max |[agg|x]| per layer (math 1 + need_backward: feeds the f16 dW scales)
    int acts_layer;   // inference (need_backward == 0): store only this layer's activations (-1: every layer's)
    // TD loss of the DQN update folded into the tail (mode 0, hexgnn_qnet_forward_td): graph g's selected node td_sel[g] (a
    // global row of THAT graph), target and importance weight -> td error, the graph's loss term, d loss / d Q of its rows
    const long long* td_sel; const float* td_tgt; const float* td_w; int td_loss_fn;
    float* td_dq; float* td_out; float* td_loss_part;
};

struct QBwdArgs {
    int n, b, H, L, mode, body_layers;
    const int* gptr; const int* rowptr_t; const int* col_t; const float* invdeg;
    const char* wpack; size_t bwd_off[kMaxL]; size_t bias_off[kMaxL];
    const float* acts;
    const float* lin_w; const float* v0_w; const float* v1_w;
    const float* adv_raw; const int* amax; const int* amin; const float* z; const float* vraw;
    const float* dq; const float* d_out_v;
    float* G; float* d_embeds;
    float* dadv; float* dz; float* dvr; float* lin_part;
    int* status;
    unsigned* gmax;   // [L] bit patterns of max |G_l| per layer (math 1)
    // raw first layer: per-graph partial weight gradient [b][17][HP] = G_0^T [agg0(8) | x0(8) | 1] (null: not produced)
    const float* x; int x_stride; int c_in; const float* agg0; float* first_part;
};

template <int NT> struct QLds {
    static constexpr int HP = 16 * NT;
    static constexpr int XS = HP + 4;                       // row stride (floats): (4NT+1) 16-B slots, odd
    static constexpr int kHalf = NT * NT * 64;              // float4 per weight half
    static constexpr int off_w = 0;
    static constexpr int off_x = 2 * kHalf * 16;
    static constexpr int off_rp = off_x + (kRows + 1) * XS * 4;   // row kRows stays all zero (gather filler)
    static constexpr int off_col = off_rp + 272;            // (kRows+2) u16, padded
    static constexpr int col_cap = (kLdsBytes - 64 - 4 * HP - off_col) < 8192 ? (kLdsBytes - 64 - 4 * HP - off_col) : 8192;
    static constexpr int off_max = off_col + col_cap;       // 8 per-wave maxima (math 1), 64 B
    static constexpr int off_bias = off_max + 64;           // the current layer's bias row (forward), HP floats
    // 16 KB of scratch (first-layer operands, head-tail reductions): aliases the weight halves when they are
    // large enough (NT >= 4), otherwise a region of its own (small widths leave plenty of LDS)
    static constexpr bool scr_alias = NT >= 4;
    static constexpr int scr_bytes = 16384;
    static constexpr int scr0_bytes = kRows * 48 * 4;   // backward: [rows][48] raw first-layer inputs (MFMA operand)
    static constexpr int off_scr_first = scr_alias ? off_w : off_bias + 4 * HP;   // half A
    static constexpr int off_scr_tail = scr_alias ? off_w : off_bias + 4 * HP;
    static constexpr int off_scr_bwd0 = scr_alias ? off_w : off_bias + 4 * HP;
    static constexpr int total = off_bias + 4 * HP + (scr_alias ? 0 : (scr0_bytes > scr_bytes ? scr0_bytes : scr_bytes));
    static_assert(col_cap >= 1024 && total <= kLdsBytes, "LDS budget");
    static_assert(!scr_alias || kHalf * 16 >= scr_bytes, "scratch must fit one weight half");
    static_assert(!scr_alias || 2 * kHalf * 16 >= scr0_bytes, "first-layer scratch must fit the weight halves");
    static_assert((1160 + 20 * HP) <= (kRows + 1) * XS, "backward head-tail scratch must fit the row buffer");
};

// sum over the 16 lanes of a DPP row (lanes 16k..16k+15), every lane receiving the total; same pairing as the xor butterfly
// 1, 2, 4, 8, on the VALU instead of through the LDS crossbar
__device__ __forceinline__ float row16_sum(float v) {
    auto dpp = [](float x, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, true));
    };
    v += dpp(v, std::integral_constant<int, 0xB1>{});     // quad_perm [1,0,3,2]
    v += dpp(v, std::integral_constant<int, 0x4E>{});     // quad_perm [2,3,0,1]
    v += dpp(v, std::integral_constant<int, 0x141>{});    // row_half_mirror
    v += dpp(v, std::integral_constant<int, 0x140>{});    // row_mirror
    return v;
}
// sum over the 8 lanes 8k..8k+7 (DPP, same scheme as row16_sum)
__device__ __forceinline__ float oct_sum(float v) {
    auto dpp = [](float x, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, true));
    };
    v += dpp(v, std::integral_constant<int, 0xB1>{});
    v += dpp(v, std::integral_constant<int, 0x4E>{});
    v += dpp(v, std::integral_constant<int, 0x141>{});    // row_half_mirror: lane i <-> 7 - i within each 8
    return v;
}
__device__ __forceinline__ float wsum64(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// CSR of the workgroup's graph -> LDS (u16 row offsets, u8 local column ids).  Returns false when it does not fit
// (the caller then walks the global CSR).
template <int NT>
__device__ __forceinline__ bool load_csr(char* lds, const int* __restrict__ rowptr, const int* __restrict__ col,
                                         int r0, int cnt, int e0, int ne, int* status) {
    using LD = QLds<NT>;
    unsigned short* s_rp = reinterpret_cast<unsigned short*>(lds + LD::off_rp);
    unsigned char* s_col = reinterpret_cast<unsigned char*>(lds + LD::off_col);
    const bool fits = ne <= LD::col_cap;
    if (!fits) return false;
    for (int i = threadIdx.x; i <= cnt; i += 512) s_rp[i] = (unsigned short)(rowptr[r0 + i] - e0);
    for (int e = threadIdx.x; e < ne; e += 512) {
        const int j = col[e0 + e] - r0;
        if (j < 0 || j >= cnt) { atomicOr(status, 4); s_col[e] = 0; }
        else s_col[e] = (unsigned char)j;
    }
    return true;
}




// global -> LDS copy of `count` float4 with all loads of a thread issued before its first LDS write
template <int kMaxPer>
__device__ __forceinline__ void copy_f4_to_lds(f32x4* __restrict__ dst, const f32x4* __restrict__ src, int count) {
    f32x4 tmp[kMaxPer];
#pragma unroll
    for (int k = 0; k < kMaxPer; ++k) { const int i = threadIdx.x + 512 * k; if (i < count) tmp[k] = src[i]; }
#pragma unroll
    for (int k = 0; k < kMaxPer; ++k) { const int i = threadIdx.x + 512 * k; if (i < count) dst[i] = tmp[k]; }
}

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// x*s = hi + lo (+ O(2^-22 |x s|)) with hi, lo in fp16; s is the row's power-of-two scale (exact).  Written pair by pair
// on 2-vectors so that hipcc emits v_cvt_pk_f16_f32 for the hi halves and v_fma_mixlo/mixhi_f16 (x*s - hi, fused, straight
// into the low / high half of the operand dword) for the lo halves instead of per-element converts and register shuffles.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2(const float x0, const float x1, const float s, unsigned& hi, unsigned& lo) {
    const float v0 = x0 * s, v1 = x1 * s;
    const f16x2 h = {(_Float16)v0, (_Float16)v1};
    const f16x2 l = {(_Float16)(v0 - (float)h[0]), (_Float16)(v1 - (float)h[1])};
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}
__device__ __forceinline__ void split_pair(const f32x4 a, const f32x4 b, const float s, f16x8& hi, f16x8& lo) {
    u32x4v H, L;
    unsigned h, l;
    split2(a[0], a[1], s, h, l); H[0] = h; L[0] = l;
    split2(a[2], a[3], s, h, l); H[1] = h; L[1] = l;
    split2(b[0], b[1], s, h, l); H[2] = h; L[2] = l;
    split2(b[2], b[3], s, h, l); H[3] = h; L[3] = l;
    hi = __builtin_bit_cast(f16x8, H);
    lo = __builtin_bit_cast(f16x8, L);
}
__device__ __forceinline__ void split_one(const f32x4 a, const float s, f16x4& hi, f16x4& lo) {
    u32x2v H, L;
    unsigned h, l;
    split2(a[0], a[1], s, h, l); H[0] = h; L[0] = l;
    split2(a[2], a[3], s, h, l); H[1] = h; L[1] = l;
    hi = __builtin_bit_cast(f16x4, H);
    lo = __builtin_bit_cast(f16x4, L);
}

__device__ __forceinline__ void row_scale(const float m, float& s, float& inv) { pow2_scale(m, s, inv); }
template <int NT>
__device__ __forceinline__ float frag_absmax(const f32x4 (&x)[NT], float m) {
#pragma unroll
    for (int c = 0; c < NT; ++c) m = fmaxf(fmaxf(m, fmaxf(fabsf(x[c][0]), fabsf(x[c][1]))), fmaxf(fabsf(x[c][2]), fabsf(x[c][3])));
    return m;
}
__device__ __forceinline__ float row_max4(float m) {   // max over the 4 lanes (l, l^16, l^32, l^48) that share a row
    m = fmaxf(m, __shfl_xor(m, 16));
    return fmaxf(m, __shfl_xor(m, 32));
}
__device__ __forceinline__ float rows_max16(float m) {  // ... then over the wave's 16 rows
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
    return m;
}

// ag[c] += rows[j][chunk c] for every neighbour j of this lane's row; the NT reads of one neighbour are issued
// together (distinct registers) and the next neighbour id is fetched one iteration ahead.
template <int NT, int XS>
__device__ __forceinline__ void gather_lds(const float* __restrict__ rows, const unsigned char* __restrict__ s_col,
                                           int eb, int ee, int g, f32x4 (&ag)[NT]) {
    int e = eb;
    // two neighbours per iteration: 2*NT reads of 16 B in flight per lane before the first add
    while (e + 1 < ee) {
        const int j0 = (int)s_col[e], j1 = (int)s_col[e + 1];
        e += 2;
        const f32x4* x0 = reinterpret_cast<const f32x4*>(rows + j0 * XS) + g;
        const f32x4* x1 = reinterpret_cast<const f32x4*>(rows + j1 * XS) + g;
        f32x4 t0[NT], t1[NT];
#pragma unroll
        for (int c = 0; c < NT; ++c) t0[c] = x0[4 * c];
#pragma unroll
        for (int c = 0; c < NT; ++c) t1[c] = x1[4 * c];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < NT; ++c) ag[c] += t0[c];      // ascending neighbour order kept: (.. + x_j0) + x_j1
#pragma unroll
        for (int c = 0; c < NT; ++c) ag[c] += t1[c];
    }
    if (e < ee) {
        const f32x4* x0 = reinterpret_cast<const f32x4*>(rows + (int)s_col[e] * XS) + g;
        f32x4 t0[NT];
#pragma unroll
        for (int c = 0; c < NT; ++c) t0[c] = x0[4 * c];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < NT; ++c) ag[c] += t0[c];
    }
}

// The neighbour lists never change between layers: every lane keeps the LDS byte offsets of its row's first eight
// neighbour rows in four registers (two u16 each, the lane's 16-byte column slot included), so a layer's gather issues
// its row reads without first fetching column ids (one dependent LDS round trip per neighbour pair less).  Slots past
// the row's degree point at the all-zero row kRows: every lane of a wave runs the same wave-uniform number of steps,
// no divergence, and x + 0 leaves the sums unchanged.
struct NbrRegs {
    unsigned off[4];    // byte offsets of neighbours 0..7 within the row buffer, + 16 g
    int eb, ee;         // edge range beyond the eighth neighbour in the LDS CSR (eb == ee: none)
    int wmax;           // wave-uniform max of min(deg, 8)
    bool wlong;         // wave-uniform: some row of the wave has more than eight neighbours
};
template <int XS>
__device__ __forceinline__ NbrRegs load_nbrs(const unsigned short* __restrict__ s_rp, const unsigned char* __restrict__ s_col,
                                             int lrow, bool rvalid, int g) {
    static_assert((kRows + 1) * XS * 4 <= 65536, "row offsets are kept as u16");
    NbrRegs nb;
    const int eb = rvalid ? (int)s_rp[lrow] : 0;
    nb.ee = rvalid ? (int)s_rp[lrow + 1] : 0;
    const int deg = nb.ee - eb;
    nb.eb = min(eb + 8, nb.ee);
#pragma unroll
    for (int k = 0; k < 4; ++k) nb.off[k] = 0u;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int j = k < deg ? (int)s_col[eb + k] : kRows;
        nb.off[k >> 1] |= (unsigned)(j * (XS * 4) + 16 * g) << (16 * (k & 1));
    }
    int m = deg;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = max(m, __shfl_xor(m, o));
    m = __builtin_amdgcn_readfirstlane(m);
    nb.wmax = min(m, 8);
    nb.wlong = m > 8;
    return nb;
}
// The same registers when the graph's CSR does not fit the LDS arrays (more than col_cap edges: never a Hex board): ids and
// bounds come from the global CSR, eb / ee stay relative to the graph's first edge e0.
template <int XS>
__device__ __forceinline__ NbrRegs load_nbrs_global(const int* __restrict__ rowptr, const int* __restrict__ col, int r0, int cnt,
                                                    int e0, int lrow, bool rvalid, int g) {
    NbrRegs nb;
    const int eb = rvalid ? rowptr[r0 + lrow] - e0 : 0;
    nb.ee = rvalid ? rowptr[r0 + lrow + 1] - e0 : 0;
    const int deg = nb.ee - eb;
    nb.eb = min(eb + 8, nb.ee);
#pragma unroll
    for (int k = 0; k < 4; ++k) nb.off[k] = 0u;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        int j = kRows;
        if (k < deg) { j = col[e0 + eb + k] - r0; if (j < 0 || j >= cnt) j = kRows; }
        nb.off[k >> 1] |= (unsigned)(j * (XS * 4) + 16 * g) << (16 * (k & 1));
    }
    int m = deg;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = max(m, __shfl_xor(m, o));
    m = __builtin_amdgcn_readfirstlane(m);
    nb.wmax = min(m, 8);
    nb.wlong = m > 8;
    return nb;
}
// neighbours beyond the eighth, ids from the global CSR (the rare path of the rare case)
template <int NT, int XS>
__device__ __forceinline__ void gather_global_tail(const float* __restrict__ rows, const int* __restrict__ col, int r0, int cnt,
                                                   int e_abs_b, int e_abs_e, int g, f32x4 (&ag)[NT]) {
    for (int e = e_abs_b; e < e_abs_e; ++e) {
        const int j = col[e] - r0;
        if (j < 0 || j >= cnt) continue;
        const f32x4* xj = reinterpret_cast<const f32x4*>(rows + j * XS) + g;
#pragma unroll
        for (int c = 0; c < NT; ++c) ag[c] += xj[4 * c];
    }
}
// ---- filler-carrying contraction ---------------------------------------------------------------------------------
// A wave's non-MFMA work issued under its PARTNER's MFMA stream runs ~3x slower (measured: gathers, epilogues, even
// vector-memory issue), but small groups of instructions placed between a wave's OWN MFMAs are nearly free.  So each K-half
// contraction carries "fillers": fill(integral_constant<q>) is called after every group of MFMAs (kGaps groups per half)
// and issues a few instructions of independent work - the LDS gather of the aggregate, global stores / loads, LDS-DMA.
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) { f(std::integral_constant<int, B>{}); static_for<B + 1, E>(f); }
}
// hipcc 7.2 lets the destination of an MFMA whose accumulator moves (vdst != srcC) overlap a source operand that dies at
// that instruction (seen at NT = 2: `v_mfma_f32_16x16x4_f32 v[12:15], v29, v15, v[34:37]`, the 4th result register wrong
// on the hardware; found by the width-24 parity test).  An empty asm that names the operands after the MFMA group keeps
// them alive across it, so the allocator cannot place a destination on them.
template <typename T> __device__ __forceinline__ void keep_alive(const T& v) { asm volatile("" :: "v"(v)); }
// Found by the width-24 parity test (NT = 2): with fewer than three accumulators in rotation an MFMA waits inside the
// matrix pipe for its srcC (the previous result on the same accumulator); when the accumulator moves (vdst != srcC) hipcc
// hands the dead srcC registers to the next LDS read (the fragment reload or a gather filler behind the group), and that
// read's data came back before the queued MFMA had read srcC: the 4th accumulator register of a tile was wrong.  Nothing
// interlocks an LDS return against a pending MFMA source, so narrow widths wait out the dependent chain (2 x 40 cycles)
// before anything else is issued; from three tiles on the accumulators rotate further apart than the MFMA latency.
__device__ __forceinline__ void mfma_drain() {
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
}
template <int NT, int MATH> struct Gaps {
    static constexpr int value = MATH == 0 ? 4 * NT : (NT / 2) * NT + (NT & 1) * NT;
};
// One K-half of a layer: acc[t] += sum_c W(c,t)^T * x[c] over the NT feature chunks of this lane's row.
//   MATH 0: exact fp32 MFMA (v_mfma_f32_16x16x4_f32), weights packed as float4 fragments; `scale` unused.
//   MATH 1: split precision ("f16x3"): W s_W = Whi + Wlo, x s = xhi + xlo in fp16 (s = this row's power-of-two scale),
//           acc += Wlo*xhi + Whi*xlo + Whi*xhi on v_mfma_f32_16x16x32_f16 (chunk pairs) / v_mfma_f32_16x16x16_f16
//           (odd last chunk), fp32 accumulate; the caller multiplies by 1/(s s_W).
//   ZC (exact fp32 only): the accumulators start from zero -- the first MFMA of every tile takes the constant 0 as its
//   srcC instead of a zero-filled register set (VALU moves cost MFMA time).
template <int NT, int MATH, bool ZC = false, typename F>
__device__ __forceinline__ void contract_half_fill(const f32x4* __restrict__ whalf, int lane, const f32x4 (&x)[NT],
                                                   f32x4 (&acc)[NT], const float scale, F&& fill) {
    // The scheduling barriers pin the fillers between the MFMA groups, so the weight fragments of the next group are
    // requested explicitly ahead of a filler instead of by the compiler's own hoisting.
    if constexpr (MATH == 0) {
        // one fragment set: the next chunk's fragments are requested into the same registers right behind the chunk's last
        // MFMA group (an MFMA reads its operands at issue); the SIMD's other wave covers the LDS round trip
        f32x4 w[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) w[t] = whalf[t * 64 + lane];
        static_for<0, NT>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            static_for<0, 4>([&](auto jj) {
                constexpr int j = decltype(jj)::value;
                // the order inside a group is pinned (a scheduling barrier behind every MFMA): the NT accumulators rotate in
                // tile order, so two MFMAs on one accumulator are exactly NT >= 3 slots apart, also across group boundaries
                // (left to itself hipcc permutes a group, and the same tile can close one group and open the next)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    if constexpr (ZC && c == 0 && j == 0) acc[t] = mfma16x16x4(w[t][j], x[c][j], f32x4{0.f, 0.f, 0.f, 0.f});
                    else acc[t] = mfma16x16x4(w[t][j], x[c][j], acc[t]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (NT < 3) mfma_drain();
                keep_alive(x[c][j]);
#pragma unroll
                for (int t = 0; t < NT; ++t) keep_alive(w[t]);
                if constexpr (j == 3 && c + 1 < NT) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) w[t] = whalf[((c + 1) * NT + t) * 64 + lane];
                }
                __builtin_amdgcn_sched_barrier(0);
                fill(std::integral_constant<int, 4 * c + j>{});
                __builtin_amdgcn_sched_barrier(0);
            });
        });
    } else {
        const char* wb = reinterpret_cast<const char*>(whalf);
        constexpr int kUnits = (NT / 2) * NT + (NT & 1) * NT;     // (chunk pair | odd last chunk) x output tile
        // unit u: pair p = u / NT (p == NT/2: the odd last chunk), tile t = u % NT
        f16x8 wh[2], wl[2];      // the odd chunk's fragments use the low halves
        auto wload = [&](auto uu, f16x8& h, f16x8& l) {
            constexpr int u = decltype(uu)::value, p = u / NT, t = u % NT;
            if constexpr (p < NT / 2) {
                const char* ub = wb + (2 * p) * NT * 1024 + lane * 16 + t * 2048;
                h = *reinterpret_cast<const f16x8*>(ub);
                l = *reinterpret_cast<const f16x8*>(ub + 1024);
            } else {
                const char* ub = wb + (NT - 1) * NT * 1024 + lane * 8 + t * 1024;
                const f16x4 h4 = *reinterpret_cast<const f16x4*>(ub);
                const f16x4 l4 = *reinterpret_cast<const f16x4*>(ub + 512);
                h = __builtin_shufflevector(h4, h4, 0, 1, 2, 3, 0, 1, 2, 3);
                l = __builtin_shufflevector(l4, l4, 0, 1, 2, 3, 0, 1, 2, 3);
            }
        };
        wload(std::integral_constant<int, 0>{}, wh[0], wl[0]);
        f16x8 xh, xl;
        // The three MFMAs of a unit form a dependent chain on one accumulator, so the last one waits inside the matrix pipe
        // (see mfma_drain): the accumulator's previous registers stay reserved for one more unit, which keeps hipcc from
        // handing them to the LDS reads that follow the unit.
        f32x4 held = acc[0];
        static_for<0, kUnits>([&](auto uu) {
            constexpr int u = decltype(uu)::value, p = u / NT, t = u % NT;
            const f32x4 before = acc[t];
            if constexpr (t == 0) {
                if constexpr (p < NT / 2) split_pair(x[2 * p], x[2 * p + 1], scale, xh, xl);
                else {
                    f16x4 h4, l4;
                    split_one(x[NT - 1], scale, h4, l4);
                    xh = __builtin_shufflevector(h4, h4, 0, 1, 2, 3, 0, 1, 2, 3);
                    xl = __builtin_shufflevector(l4, l4, 0, 1, 2, 3, 0, 1, 2, 3);
                }
            }
            if constexpr (u + 1 < kUnits) wload(std::integral_constant<int, u + 1>{}, wh[(u + 1) & 1], wl[(u + 1) & 1]);
            if constexpr (p < NT / 2) {
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[u & 1], xh, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[u & 1], xl, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[u & 1], xh, acc[t], 0, 0, 0);
            } else {
                const f16x4 h4 = __builtin_shufflevector(wh[u & 1], wh[u & 1], 0, 1, 2, 3);
                const f16x4 l4 = __builtin_shufflevector(wl[u & 1], wl[u & 1], 0, 1, 2, 3);
                const f16x4 xh4 = __builtin_shufflevector(xh, xh, 0, 1, 2, 3), xl4 = __builtin_shufflevector(xl, xl, 0, 1, 2, 3);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x16f16(l4, xh4, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x16f16(h4, xl4, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x16f16(h4, xh4, acc[t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            keep_alive(wh[u & 1]); keep_alive(wl[u & 1]); keep_alive(xh); keep_alive(xl);
            keep_alive(held);
            held = before;
            if constexpr (NT < 3) mfma_drain();      // (wider kernels: load_guard() inside the fillers that issue loads)
            fill(std::integral_constant<int, u>{});
            __builtin_amdgcn_sched_barrier(0);
        });
    }
}

// The gather of the first eight neighbours as 32 micro-ops spread over the G gaps of a contraction: for neighbour k the
// read of its row chunks into ONE landing buffer, then three add steps (thirds of the chunks); the MFMA group between two
// gaps covers the LDS latency.  Ascending order of the sums is kept.
template <int NT>
__device__ __forceinline__ void gather_read(const char* base, unsigned off, f32x4 (&t)[NT]) {
    const f32x4* x = reinterpret_cast<const f32x4*>(base + off);
#pragma unroll
    for (int c = 0; c < NT; ++c) t[c] = x[4 * c];
}
// A filler that issues a load right behind a split-precision unit first gives the unit's dependent MFMA chain 48 cycles to
// start its last link (see mfma_drain; exact fp32 rotates 3+ accumulators and needs no guard).
template <int MATH> __device__ __forceinline__ void load_guard() {
    if constexpr (MATH == 1) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
}
// Landing buffers of the gather: one at the wide widths (the kernels sit at the register ceiling there), three in rotation
// at 33..64 columns with exact fp32 math, where registers are plentiful and a phase is short: neighbour k is read in gap k and
// added in gap k + 2, two MFMA groups later.  (With one buffer and only 4 NT = 12 gaps for 32 micro-ops a neighbour's read and
// its first add fell into the SAME gap: the LDS round trip was exposed eight times per layer -- phase S of GNN-S took 3.4 k
// ticks against 1.7 k for phase A, tools/stamps.py S256.)  The sums keep their ascending neighbour order: same bits.
template <int NT, int MATH> struct GatherLand {
    static constexpr int value = (MATH == 0 && (NT == 3 || NT == 4)) ? 3 : 1;
};
// Round 4: v_mfma_f32_16x16x4_f32 does not overlap VALU work on its SIMD (every VALU instruction adds ~3 cycles to the MFMA
// stream: tools/microbench/mfma_valu_overlap.hip), so the gather is kept to the adds it needs: neighbour 0 lands in the
// sums directly (no zero fill, no `0 + x` add), and an empty asm in every wave-uniform `if (k < wmax)` block keeps hipcc from
// turning it into "add, then select per element" (it did for a third of the adds: 92 v_cndmask per layer).
__device__ __forceinline__ void no_ifcvt() { asm volatile("" ::: "memory"); }
// max(x, 0) as ONE instruction: on an MFMA result fmaxf() costs two (hipcc canonicalises a value it cannot prove quiet first)
__device__ __forceinline__ float relu_raw(float x) { float r; asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x)); return r; }
template <int NT, int Q, int G, int MATH, int KL>
__device__ __forceinline__ void gather_gap(const float* rows, const NbrRegs& nb, f32x4 (&ag)[NT], f32x4 (&tbs)[KL][NT]) {
    const char* base = reinterpret_cast<const char*>(rows);
    if constexpr (KL == 3) {
        static_assert(G >= 10, "eight reads + two gaps of distance");
        if constexpr (Q >= 2 && Q < 10) {
            constexpr int k = Q - 2;
            if (k < nb.wmax) {       // wave-uniform
                no_ifcvt();
#pragma unroll
                for (int c = 0; c < NT; ++c) {
                    if constexpr (k == 0) ag[c] = tbs[0][c];
                    else ag[c] += tbs[k % 3][c];
                }
            } else if constexpr (k == 0) {
#pragma unroll
                for (int c = 0; c < NT; ++c) ag[c] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        if constexpr (Q < 8) {
            constexpr int k = Q;
            if (k < nb.wmax) {
                no_ifcvt();
                const unsigned o = nb.off[k >> 1];
                gather_read<NT>(base, (k & 1) ? (o >> 16) : (o & 0xffffu), tbs[k % 3]);
            }
        }
        return;
    }
    f32x4 (&tb)[NT] = tbs[0];
    static_for<0, 32>([&](auto mm) {
        constexpr int m = decltype(mm)::value;
        // 28 gaps (width 97..112, the GNN-L case): shifted by 4/32 so that a neighbour's read and its first add never share
        // a gap and an MFMA group covers the LDS latency (other widths: the shift only moved hipcc into spilling)
        constexpr int kShift = G == 28 ? 4 : 0;
        constexpr int gap = (m * G + kShift) / 32;
        if constexpr (gap == Q) {
            constexpr int k = m / 4, part = m % 4;
            if constexpr (k == 0) {
                // neighbour 0 (a row without neighbours reads the all-zero row): straight into the sums, no adds
                if constexpr (part == 0) {
                    if (0 < nb.wmax) {
                        no_ifcvt();
                        load_guard<MATH>();
                        gather_read<NT>(base, nb.off[0] & 0xffffu, ag);
                    } else {
#pragma unroll
                        for (int c = 0; c < NT; ++c) ag[c] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
            } else if (k < nb.wmax) {       // wave-uniform
                no_ifcvt();
                if constexpr (part == 0) {
                    load_guard<MATH>();
                    const unsigned o = nb.off[k >> 1];
                    gather_read<NT>(base, (k & 1) ? (o >> 16) : (o & 0xffffu), tb);
                } else {
                    constexpr int c0 = (part - 1) * NT / 3, c1 = part * NT / 3;
#pragma unroll
                    for (int c = c0; c < c1; ++c) ag[c] += tb[c];
                }
            }
        }
    });
}

// piece index of this wave's q-th share of a weight half (NT*NT pieces over 8 waves); -1: none.
// `spare` (workgroup-uniform): the graph has at most 64 rows, waves 4-7 own no rows and share the SIMDs of waves 0-3 -- they
// then move ALL the pieces (up to two rounds of four inside the kDma filler slots), and the waves with rows issue none (a piece
// costs its issuing wave 60-150 cycles of SALU + vector-memory issue, and at one active wave per SIMD nothing hides them).
// (Round 4 also tried handing waves 4-7 the whole non-MFMA side of their partner's layer -- gather, stores, DMA -- through the
// unused upper half of the row buffer: GNN-S 5.05 k -> 5.9 k ticks per layer.  A helper's VALU instructions issue only between
// its partner's fp32 MFMAs, one per 32-cycle slot: profiles/r04/helper_waves_experiment.patch, stamps_S256_helper_waves.txt.)
template <int NT> __device__ __forceinline__ int dma_share(int wave, int q, bool spare = false) {
    if (spare) {
        if (wave < 4) return -1;
        const int p0 = (wave - 4) + 8 * q;                   // two pieces per slot and spare wave: the second is dma_share2's
        return p0 < NT * NT ? p0 : -1;
    }
    const int p = wave + 8 * q;
    return p < NT * NT ? p : -1;
}
// the second piece of a slot in `spare` mode (-1: none)
template <int NT> __device__ __forceinline__ int dma_share2(int wave, int q, bool spare) {
    if (!spare || wave < 4) return -1;
    const int p = (wave - 4) + 8 * q + 4;
    return p < NT * NT ? p : -1;
}

// ================================================= forward =================================================
template <int NT, int MATH>
__global__ __launch_bounds__(512) void qnet_fwd_kernel(QFwdArgs a) {
    using LD = QLds<NT>;
    constexpr int HP = LD::HP, XS = LD::XS, kHalf = LD::kHalf;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    f32x4* wbuf = reinterpret_cast<f32x4*>(lds + LD::off_w);       // [2][kHalf]
    float* xbuf = reinterpret_cast<float*>(lds + LD::off_x);       // [kRows][XS]
    const unsigned short* s_rp = reinterpret_cast<const unsigned short*>(lds + LD::off_rp);
    const unsigned char* s_col = reinterpret_cast<const unsigned char*>(lds + LD::off_col);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int gi = blockIdx.x;
    QSTAMP(0, 0, 0);
    // Requested before anything that depends on the graph (round 4): W_r of the first hidden layer (half B) and the raw first
    // layer's weights -- their round trips overlap the chain gptr -> rowptr -> columns instead of following it.
    constexpr int kStage1 = (LD::kHalf + 511) / 512, kStage0 = (2 * HP * kSmallCin / 4 + 511) / 512;
    f32x4 wstg1[kStage1], wstg0[kStage0];
    {
        const f32x4* src1 = reinterpret_cast<const f32x4*>(a.wpack + a.fwd_off[a.L > 1 ? 1 : 0]) + kHalf;
#pragma unroll
        for (int k = 0; k < kStage1; ++k) { const int i = tid + 512 * k; if (a.L > 1 && i < kHalf) wstg1[k] = src1[i]; }
        const f32x4* src0 = reinterpret_cast<const f32x4*>(a.wpack + a.fwd_off[0]);
#pragma unroll
        for (int k = 0; k < kStage0; ++k) { const int i = tid + 512 * k; if (i < 2 * HP * kSmallCin / 4) wstg0[k] = src0[i]; }
    }
    const int r0 = a.gptr[gi], r1 = a.gptr[gi + 1];
    const int cnt = r1 - r0;
    if (cnt > kRows) {
        // a graph that does not fit the tile reached this kernel (stale size hint): flag it AND poison its outputs, so
        // the failure is visible in the data even if nobody reads the status word
        if (tid == 0) {
            atomicOr(a.status, 2);
            if (a.out_v) a.out_v[gi] = __builtin_nanf("");
            if (a.td_sel && a.mode == 0) { a.td_out[gi] = __builtin_nanf(""); a.td_loss_part[gi] = __builtin_nanf(""); }
        }
        for (int i = tid; i < cnt; i += 512) {
            a.q[r0 + i] = __builtin_nanf("");
            if (a.td_sel && a.mode == 0) a.td_dq[r0 + i] = __builtin_nanf("");
        }
        return;
    }
    const int H = a.H;
    const int lrow = wave * 16 + r;                 // local row of this lane
    const bool rvalid = lrow < cnt;
    const bool wactive = wave * 16 < cnt;           // wave-uniform
    const bool spare = cnt <= kRows / 2;            // workgroup-uniform: waves 4-7 own no rows (dma_share)
    const int grow = r0 + lrow;
    // values the prologue needs two or three barriers further down are requested NOW, with the CSR: every global round trip
    // left on the chain gptr -> rowptr -> columns -> ... costs ~0.8 us at kernel start (GNN-S: 12.5 k ticks of prologue)
    const float idg = rvalid ? a.invdeg[grow] : 0.f;                 // hidden layers: 1 / deg of this lane's row
    const float sc0 = tid < cnt ? a.invdeg[r0 + tid] : 0.f;          // raw first layer: thread tid sums row tid
    // (the first layer's bias row too, at the narrow widths: at 97..112 columns its 28 registers cost the layer loop's
    // allocation 1.4 us, measured, and the bias is then read where it is used)
    constexpr bool kBiasAhead = NT <= 4;
    f32x4 b0v[kBiasAhead ? NT : 1];
    if constexpr (kBiasAhead) {
        const f32x4* b0 = reinterpret_cast<const f32x4*>(a.wpack + a.bias_off[0]);
#pragma unroll
        for (int t = 0; t < NT; ++t) b0v[t] = b0[4 * t + g];
    }
    // (exact fp32: the first hidden layer's bias row, which its accumulators start from)
    f32x4 b1stg = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (MATH == 0) { if (a.L > 1 && tid < HP / 4) b1stg = reinterpret_cast<const f32x4*>(a.wpack + a.bias_off[1])[tid]; }
    const int e0 = a.rowptr[r0], ne = a.rowptr[r1] - e0;
    const bool csr_lds = load_csr<NT>(lds, a.rowptr, a.col, r0, cnt, e0, ne, a.status);
    float* s_max = reinterpret_cast<float*>(lds + LD::off_max);      // per-wave maxima (math 1)
    if (tid < 16) s_max[tid] = 0.f;
    if (tid < XS) xbuf[kRows * XS + tid] = 0.f;                      // the gather's filler row

    // ---- stage W_r of layer 1 into half B (the self half runs first); first-layer scratch lives in half A ----
    if (a.L > 1) {
#pragma unroll
        for (int k = 0; k < kStage1; ++k) { const int i = tid + 512 * k; if (i < kHalf) wbuf[kHalf + i] = wstg1[k]; }
    }
    NbrRegs nbr;
    float* s_w0 = reinterpret_cast<float*>(lds + LD::off_scr_first);  // [2][HP][8]
    float* s_f = s_w0 + 2 * HP * kSmallCin;                          // [kRows][16]: agg0 | x0
    {
        // raw features of the graph's rows -> LDS (x0 half of s_f), first-layer weights -> LDS: all independent
        // global loads, one barrier; the neighbour sums then run on LDS only.
#pragma unroll
        for (int k = 0; k < kStage0; ++k) {
            const int i = tid + 512 * k;
            if (i < 2 * HP * kSmallCin / 4) reinterpret_cast<f32x4*>(s_w0)[i] = wstg0[k];
        }
#pragma unroll
        for (int i = tid; i < kRows * kSmallCin; i += 512) {
            const int rr = i / kSmallCin, qq = i % kSmallCin;
            s_f[rr * 16 + 8 + qq] = (rr < cnt && qq < a.c_in) ? a.x[(size_t)(r0 + rr) * a.x_stride + qq] : 0.f;
        }
        __syncthreads();
        nbr = csr_lds ? load_nbrs<XS>(s_rp, s_col, lrow, rvalid, g)
                      : load_nbrs_global<XS>(a.rowptr, a.col, r0, cnt, e0, lrow, rvalid, g);
        if (tid < kRows) {
            float ag0[kSmallCin];
#pragma unroll
            for (int qq = 0; qq < kSmallCin; ++qq) ag0[qq] = 0.f;
            if (tid < cnt) {
                const int row = r0 + tid;
                if (csr_lds) {
                    for (int e = s_rp[tid]; e < s_rp[tid + 1]; ++e) {
                        const float* xr = s_f + (int)s_col[e] * 16 + 8;
#pragma unroll
                        for (int qq = 0; qq < kSmallCin; ++qq) ag0[qq] += xr[qq];
                    }
                } else {
                    for (int e = a.rowptr[row]; e < a.rowptr[row + 1]; ++e) {
                        const float* xr = s_f + (a.col[e] - r0) * 16 + 8;
#pragma unroll
                        for (int qq = 0; qq < kSmallCin; ++qq) ag0[qq] += xr[qq];
                    }
                }
#pragma unroll
                for (int qq = 0; qq < kSmallCin; ++qq) ag0[qq] *= sc0;
                if (a.need_backward) {
                    f32x4* ao = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.saved + a.agg_off[0]) + (size_t)row * kSmallCin);
                    ao[0] = f32x4{ag0[0], ag0[1], ag0[2], ag0[3]};
                    ao[1] = f32x4{ag0[4], ag0[5], ag0[6], ag0[7]};
                }
            }
#pragma unroll
            for (int qq = 0; qq < kSmallCin; ++qq) s_f[tid * 16 + qq] = ag0[qq];
        }
    }
    __syncthreads();

    // ---- layer 0 (raw features): every lane produces its own row chunks, already in the chained layout ----
    f32x4 xs[NT];
    {
        float f[16];
#pragma unroll
        for (int qq = 0; qq < 16; ++qq) f[qq] = s_f[lrow * 16 + qq];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            f32x4 v;
            if constexpr (kBiasAhead) v = b0v[t];
            else v = reinterpret_cast<const f32x4*>(a.wpack + a.bias_off[0])[4 * t + g];
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int o = 16 * t + 4 * g + q4;
                const float* wl0 = s_w0 + o * kSmallCin;
                const float* wr0 = s_w0 + HP * kSmallCin + o * kSmallCin;
                float s = v[q4];
                if (a.c_in <= 2) {   // the model's raw features [degree, is_terminal]: the remaining packed weights are zero
                    s += wl0[0] * f[0] + wr0[0] * f[8];
                    s += wl0[1] * f[1] + wr0[1] * f[9];
                } else {
#pragma unroll
                    for (int qq = 0; qq < kSmallCin; ++qq) s += wl0[qq] * f[qq] + wr0[qq] * f[8 + qq];
                }
                v[q4] = rvalid ? fmaxf(s, 0.f) : 0.f;
            }
            xs[t] = v;
        }
        f32x4* xr = reinterpret_cast<f32x4*>(xbuf + lrow * XS) + g;
#pragma unroll
        for (int t = 0; t < NT; ++t) xr[4 * t] = xs[t];
        // (layer 0's rows go to global memory with the first hidden layer's fillers, like every other layer's: a store
        // here would be waited for at the barrier below)
    }
    if constexpr (MATH == 0) { if (tid < HP / 4) reinterpret_cast<f32x4*>(lds + LD::off_bias)[tid] = b1stg; }
    __syncthreads();   // xbuf + half B (+ the first hidden layer's bias row) visible; half A (scratch) free

    // ---- hidden layers ----
    const size_t slab = (size_t)a.n * HP;
    const float validf = rvalid ? 1.f : 0.f;
    QSTAMP(0, 0, 1);
    // Per layer two phases, each a K-half contraction that carries the layer's other work as fillers between its MFMAs:
    //   phase S: self half (W_r, half B) on the rows kept in registers; fillers = the LDS gather of the aggregate, the
    //            previous layer's saved-activation stores, LDS-DMA of W_l(l) into half A           -> barrier 1
    //   phase A: aggregate half (W_l, half A); fillers = the aggregate's saved-tensor stores, LDS-DMA of W_r(l+1) into
    //            half B; then bias + ReLU + new rows to LDS                                        -> barrier 2
    constexpr int kGaps = Gaps<NT, MATH>::value;
    constexpr int kDma = (NT * NT + 7) / 8;                  // LDS-DMA pieces per wave and half
    constexpr int kFill = kDma + NT;                         // filler slots used per phase
    constexpr int kTail = kFill > kGaps ? kGaps : kFill;     // narrow widths: the slots past the last gap run after the MFMAs
    float* s_bias = reinterpret_cast<float*>(lds + LD::off_bias);
    // where this lane's accumulators start (exact fp32): the bias row, or the all-zero row for a pad row
    const f32x4* binit = reinterpret_cast<const f32x4*>(rvalid ? s_bias : xbuf + kRows * XS) + g;
    const unsigned lds_w = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(lds + LD::off_w);
    const unsigned rowoff = (unsigned)grow * (HP * 4) + 16 * g;      // byte offset of this lane's slot inside a [n][HP] slab
    const unsigned lane16 = 16 * lane;
    auto acts_off = [&](const int l) -> unsigned {        // lane offset for storing layer l's rows (kOob: not stored)
        return (rvalid && (a.need_backward || a.acts_layer < 0 || a.acts_layer == l)) ? rowoff : kOob;
    };
    auto publish_xmax = [&](const int l) {   // layer maximum of [agg | x] over this graph -> global (order-independent);
        if constexpr (MATH == 1) {           // called one barrier after the waves wrote s_max
            if (a.xmax && tid == 0) {
                float mm = 0.f;
#pragma unroll
                for (int w8 = 0; w8 < 8; ++w8) mm = fmaxf(mm, s_max[w8]);
                atomicMax(a.xmax + l, __builtin_bit_cast(unsigned, mm));
            }
        }
    };
    for (int l = 1; l < a.L; ++l) {
        QSTAMP(0, l, 0);
        if (l > 1) publish_xmax(l - 1);
        const f32x4* wsrc = reinterpret_cast<const f32x4*>(a.wpack + a.fwd_off[l]);
        const bool more = l + 1 < a.L;
        const f32x4* nsrc = reinterpret_cast<const f32x4*>(a.wpack + a.fwd_off[more ? l + 1 : l]) + kHalf;
        // exact fp32: the NEXT layer's bias row is staged (its accumulators start from it); split math: this layer's
        f32x4 bstg = f32x4{0.f, 0.f, 0.f, 0.f};
        if (tid < HP / 4) bstg = reinterpret_cast<const f32x4*>(a.wpack + a.bias_off[(MATH == 0 && more) ? l + 1 : l])[tid];
        auto dmaS = [&](auto qq) {
            const int p = dma_share<NT>(wave, decltype(qq)::value, spare);
            if (p >= 0) dma_piece(wsrc + p * 64, lane16, lds_w + p * 1024);
            const int p2 = dma_share2<NT>(wave, decltype(qq)::value, spare);
            if (p2 >= 0) dma_piece(wsrc + p2 * 64, lane16, lds_w + p2 * 1024);
        };
        auto dmaA = [&](auto qq) {
            const int p = dma_share<NT>(wave, decltype(qq)::value, spare);
            if (more && p >= 0) dma_piece(nsrc + p * 64, lane16, lds_w + (kHalf + p * 64) * 16);
            const int p2 = dma_share2<NT>(wave, decltype(qq)::value, spare);
            if (more && p2 >= 0) dma_piece(nsrc + p2 * 64, lane16, lds_w + (kHalf + p2 * 64) * 16);
        };
        if (!wactive) {
            // a wave without rows only moves its weight pieces (its own straight path: the active path below then has no
            // `if (wactive)` regions whose merges cost register copies -- VALU instructions are MFMA time here)
            static_for<0, kDma>(dmaS);
            wait_vmem();
            if constexpr (MATH == 1) { if (tid < HP / 4) reinterpret_cast<f32x4*>(s_bias)[tid] = bstg; }
            lds_barrier();
            if constexpr (MATH == 0) { if (tid < HP / 4) reinterpret_cast<f32x4*>(s_bias)[tid] = bstg; }
            static_for<0, kDma>(dmaA);
            wait_vmem();
            lds_barrier();
            continue;
        }
        f32x4 acc[NT], ag[NT];
        if constexpr (MATH == 0) {
            // the accumulators start from the bias row (an LDS read instead of 14 zero moves + 14 adds in the epilogue);
            // pad rows start from the all-zero row and stay exactly zero through the ReLU
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = binit[4 * t];
        } else {
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        float rs = 1.f, rinv = 1.f, mx = 0.f;
        f32x4 tb[GatherLand<NT, MATH>::value][NT];
        if constexpr (MATH == 1) { mx = row_max4(frag_absmax<NT>(xs, 0.f)); row_scale(mx, rs, rinv); }
        // ---- phase S ----
        const __amdgpu_buffer_rsrc_t yprev = slab_rsrc(a.acts + slab * (l - 1));
        const unsigned yprev_off = acts_off(l - 1);
        auto fillS = [&](auto qq) {
            constexpr int Q = decltype(qq)::value;
            gather_gap<NT, Q, kGaps, MATH>(xbuf, nbr, ag, tb);
            if constexpr (Q < kDma) dmaS(qq);
            else if constexpr (Q < kDma + NT) buf_store(xs[Q - kDma], yprev, yprev_off + 64 * (Q - kDma));
        };
        contract_half_fill<NT, MATH>(wbuf + kHalf, lane, xs, acc, rs, fillS);
        static_for<kTail, kFill>(fillS);
        if (nbr.wlong) {      // (wave-uniform; pad rows: eb == ee)
            if (csr_lds) gather_lds<NT, XS>(xbuf, s_col, nbr.eb, nbr.ee, g, ag);
            else gather_global_tail<NT, XS>(xbuf, a.col, r0, cnt, e0 + nbr.eb, e0 + nbr.ee, g, ag);
        }
#pragma unroll
        for (int c = 0; c < NT; ++c) ag[c] *= idg;          // idg == 0 on pad rows
        QSTAMP(0, l, 1);
        wait_vmem();
        if constexpr (MATH == 1) { if (tid < HP / 4) reinterpret_cast<f32x4*>(s_bias)[tid] = bstg; }
        QSTAMP(0, l, 3);
        lds_barrier();     // barrier 1: half A = W_l(l) (+ bias row, split math); every gather of this layer is done; half B is free
        // (exact fp32: every wave has read this layer's bias row into its accumulators by now: the next layer's may land)
        if constexpr (MATH == 0) { if (tid < HP / 4) reinterpret_cast<f32x4*>(s_bias)[tid] = bstg; }
        QSTAMP(0, l, 4);
        // ---- phase A ----
        float rsa = 1.f;
        if constexpr (MATH == 1) {
            // the aggregate gets its own power-of-two row scale (it was not known when the self half ran); the self
            // half's sums are carried over by the exact ratio of the two scales
            float ma = row_max4(frag_absmax<NT>(ag, 0.f));
            if (a.xmax) { const float wm = rows_max16(fmaxf(ma, mx)); if (lane == 0) s_max[wave] = wm; }
            ma = fmaxf(ma, mx * 0x1p-40f);
            float rinva;
            row_scale(ma, rsa, rinva);
            const float carry = rsa * rinv;
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] *= carry;
            rinv = rinva * (reinterpret_cast<const float*>(a.wpack + a.bias_off[l]) + HP)[1];
        }
        const __amdgpu_buffer_rsrc_t ao = slab_rsrc(a.saved + a.agg_off[l]);
        const unsigned ao_off = (rvalid && a.need_backward) ? rowoff : kOob;
        auto fillA = [&](auto qq) {
            constexpr int Q = decltype(qq)::value;
            if constexpr (Q < kDma) dmaA(qq);
            else if constexpr (Q < kDma + NT) buf_store(ag[Q - kDma], ao, ao_off + 64 * (Q - kDma));
        };
        contract_half_fill<NT, MATH>(wbuf, lane, ag, acc, rsa, fillA);
        static_for<kTail, kFill>(fillA);
        QSTAMP(0, l, 5);
        {   // epilogue: (bias,) ReLU, new rows -> registers and LDS
            f32x4* xr = reinterpret_cast<f32x4*>(xbuf + lrow * XS) + g;
            const f32x4* bl = reinterpret_cast<const f32x4*>(s_bias) + g;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                f32x4 v = acc[t];
                if constexpr (MATH == 1) { v *= rinv; v += bl[4 * t]; }
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) v[q4] = relu_raw(v[q4]);
                if constexpr (MATH == 1) v *= validf;              // pad rows stay exactly zero
                xs[t] = v;
                xr[4 * t] = v;
            }
        }
        wait_vmem();
        QSTAMP(0, l, 6);
        lds_barrier();     // barrier 2: new rows + half B = W_r(l+1) visible; half A free
        QSTAMP(0, l, 7);
    }
    {   // the last layer's rows (the only layer when L == 1)
        if (a.L > 1) publish_xmax(a.L - 1);
        const __amdgpu_buffer_rsrc_t ylast = slab_rsrc(a.acts + slab * (a.L - 1));
        const unsigned ylast_off = acts_off(a.L - 1);
#pragma unroll
        for (int t = 0; t < NT; ++t) buf_store(xs[t], ylast, ylast_off + 64 * t);
    }

    // ---- head tail (scratch aliases the weight halves, free after the last barrier) ----
    float* sc = reinterpret_cast<float*>(lds + LD::off_scr_tail);
    float* s_w = sc;                 // [128] advantage weights
    float* s_pool = sc + 128;        // [4*128]
    float* s_z = sc + 640;           // [64]
    float* s_red = sc + 704;         // [8]
    float* s_misc = sc + 712;        // [0] = tanh(v)
    float* s_mx = sc + 768;          // [3][128]
    float* s_mn = sc + 1152;         // [3][128]
    float* s_sm = sc + 1536;         // [3][128]
    int* s_ax = reinterpret_cast<int*>(sc + 1920);   // [3][128]
    int* s_an = reinterpret_cast<int*>(sc + 2304);   // [3][128]
    const int H2 = H / 2, H4 = 4 * H;
    if (tid < 128) s_w[tid] = tid < H ? a.lin_w[tid] : 0.f;
    // the tail's small global operands are requested here, one round trip for all of them, instead of one each at the point
    // of use (four exposed round trips on a path with no other work to hide them)
    const float lin_b0 = a.lin_b[0];
    const int vk_ = tid >> 3;
    const float v0b_k = (a.mode != 2 && vk_ < H2) ? a.v0_b[vk_] : 0.f;
    const float v1w_l = (a.mode != 2 && wave == 0 && lane < H2) ? a.v1_w[lane] : 0.f;
    const float v1b_0 = a.mode != 2 ? a.v1_b[0] : 0.f;
    const bool td_on = a.td_sel != nullptr && a.mode == 0;
    const long long td_s = td_on ? a.td_sel[gi] : -1;
    const float td_t = td_on ? a.td_tgt[gi] : 0.f;
    const float td_wg = (td_on && a.td_w) ? a.td_w[gi] : 1.f;
    __syncthreads();
    // advantages from the registers: partial dot over this lane's chunks, reduce over the 4 lanes of the row
    float adv = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const f32x4 w = reinterpret_cast<const f32x4*>(s_w)[4 * t + g];
        adv += xs[t][0] * w[0] + xs[t][1] * w[1] + xs[t][2] * w[2] + xs[t][3] * w[3];
    }
    adv += __shfl_xor(adv, 16);
    adv += __shfl_xor(adv, 32);
    adv += lin_b0;
    const float tadv = 2.f * tanhf(adv);
    if (g == 0 && rvalid) {
        a.adv_raw[grow] = adv;
        if (a.mode == 2) a.q[grow] = tadv;
    }
    if (a.mode == 2) return;
    {   // sum of 2tanh(adv) over the graph: lanes g==0 of valid rows; fixed-shape tree
        float v = (g == 0 && rvalid) ? tadv : 0.f;
        v = wsum64(v);
        if (lane == 0) s_red[wave] = v;
    }
    // value-MLP weights -> registers now (the loads fly during pooling): thread (k = tid / 8, part = tid % 8) takes hidden
    // unit k (up to 64) and the 16-byte column groups part, part + 8, ... of its 4H-wide row (H groups, H <= 112: 14 loads)
    constexpr int kVQ = 14;
    const int vk = tid >> 3, vpart = tid & 7;
    f32x4 wv[kVQ];
    {
        const f32x4* wrow = reinterpret_cast<const f32x4*>(a.v0_w + (size_t)vk * H4);
#pragma unroll
        for (int j = 0; j < kVQ; ++j) {
            const int q = vpart + 8 * j;
            wv[j] = (vk < H2 && q < H) ? wrow[q] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    // pooling straight from the LDS rows: column c = tid&127, four row phases
    {
        const int c = tid & 127, ph = tid >> 7;
        float sum = 0.f, mx = -INFINITY, mn = INFINITY;
        int ax = -1, an = -1;
        if (c < H) {
            for (int row = ph; row < cnt; row += 4) {
                const float v = xbuf[row * XS + c];
                sum += v;
                if (v > mx) { mx = v; ax = row; }
                if (v < mn) { mn = v; an = row; }
            }
        }
        if (ph > 0) {
            const int o = (ph - 1) * 128 + c;
            s_sm[o] = sum; s_mx[o] = mx; s_mn[o] = mn; s_ax[o] = ax; s_an[o] = an;
        }
        __syncthreads();
        if (ph == 0 && c < H) {
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const int o = p * 128 + c;
                sum += s_sm[o];
                const float mx1 = s_mx[o], mn1 = s_mn[o];
                const int ax1 = s_ax[o], an1 = s_an[o];
                if (ax1 >= 0 && (ax < 0 || mx1 > mx || (mx1 == mx && ax1 < ax))) { mx = mx1; ax = ax1; }
                if (an1 >= 0 && (an < 0 || mn1 < mn || (mn1 == mn && an1 < an))) { mn = mn1; an = an1; }
            }
            if (cnt == 0) { mx = 0.f; mn = 0.f; }
            const float mean = sum / (float)max(cnt, 1);
            s_pool[c] = sum; s_pool[H + c] = mx; s_pool[2 * H + c] = mn; s_pool[3 * H + c] = mean;
            float* pg = a.pooled + (size_t)gi * H4;
            pg[c] = sum; pg[H + c] = mx; pg[2 * H + c] = mn; pg[3 * H + c] = mean;
            a.amax[(size_t)gi * H + c] = ax >= 0 ? r0 + ax : -1;
            a.amin[(size_t)gi * H + c] = an >= 0 ? r0 + an : -1;
        }
    }
    __syncthreads();
    {
        f32x4 p4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < kVQ; ++j) {
            const int q = vpart + 8 * j;
            if (q < H) p4 += wv[j] * reinterpret_cast<const f32x4*>(s_pool)[q];
        }
        float p = (p4[0] + p4[1]) + (p4[2] + p4[3]);
        p = oct_sum(p);          // over the 8 lanes that share the hidden unit
        if (vpart == 0 && vk < H2) {
            const float zz = fmaxf(p + v0b_k, 0.f);
            s_z[vk] = zz;
            a.z[(size_t)gi * H2 + vk] = zz;
        }
    }
    __syncthreads();
    if (wave == 0) {
        float p = lane < H2 ? v1w_l * s_z[lane] : 0.f;
        p = wsum64(p);
        if (lane == 0) {
            const float v = p + v1b_0;
            a.vraw[gi] = v;
            s_misc[0] = tanhf(v);
        }
    }
    __syncthreads();
    float adv_total = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) adv_total += s_red[w];
    const float mean_adv = adv_total / (float)max(cnt, 1);
    const float V = s_misc[0];
    if (a.mode == 1 && tid == 0) a.out_v[gi] = V;
    const float qv = (a.mode == 0 ? V : 0.f) + tadv - mean_adv;
    if (g == 0 && rvalid) a.q[grow] = qv;
    if (td_on) {
        // loss = mean_g w_g l(Q[sel_g] - target_g): the same per-entry expressions as td_loss_fused_kernel (head.hip), so dq
        // and td have its bits; the mean over the graphs is summed from td_loss_part by the backward's reduce launch
        if (g == 0 && rvalid) {
            float d = 0.f;
            if ((long long)grow == td_s) {
                const float e = qv - td_t, ae = fabsf(e);
                a.td_out[gi] = e;
                a.td_loss_part[gi] = td_wg * (a.td_loss_fn == 0 ? e * e : (ae <= 1.f ? 0.5f * e * e : ae - 0.5f));
                const float dl = a.td_loss_fn == 0 ? 2.f * e : fminf(fmaxf(e, -1.f), 1.f);
                d = (1.f / (float)a.b) * td_wg * dl;
            }
            a.td_dq[grow] = d;
        }
        if (tid == 0 && (td_s < (long long)r0 || td_s >= (long long)r1)) {
            // the selected node is not a row of this graph (contract of the fused form): flag AND poison
            atomicOr(a.status, 16);
            a.td_out[gi] = __builtin_nanf("");
            a.td_loss_part[gi] = __builtin_nanf("");
        }
    }
    QSTAMP(0, 0, 2);
}

// ================================================= backward =================================================
// Data-gradient chain of the whole network for one graph: head tail backward, then per layer
//   G_l = (dXs_{l+1} + sum_{j in T(i)} dAggS_{l+1,j}) * [y_l > 0];  [dAggS_l | dXs_l] = G_l [W_l | W_r]
// with dAggS rows exchanged through LDS and dXs / G kept in registers.  Writes G_l (all layers) for the batched
// weight-gradient GEMM, and the head's per-graph partials.
template <int NT, int MATH>
__global__ __launch_bounds__(512) void qnet_bwd_kernel(QBwdArgs a) {
    using LD = QLds<NT>;
    constexpr int HP = LD::HP, XS = LD::XS, kHalf = LD::kHalf;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    f32x4* wbuf = reinterpret_cast<f32x4*>(lds + LD::off_w);
    float* dbuf = reinterpret_cast<float*>(lds + LD::off_x);
    const unsigned short* s_rp = reinterpret_cast<const unsigned short*>(lds + LD::off_rp);
    const unsigned char* s_col = reinterpret_cast<const unsigned char*>(lds + LD::off_col);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int gi = blockIdx.x;
    QSTAMP(1, 0, 0);
    const int H = a.H, L = a.L;
    // Requested before anything that depends on the graph (round 4): the top layer's W_r part (staged into half B below) and this
    // thread's share of the value MLP's first-layer weights (d pooled = v0_w^T dz, three barriers further down) -- their round
    // trips then overlap the chain gptr -> rowptr -> columns instead of following it.
    constexpr int kStage = (kHalf + 511) / 512;
    f32x4 wstg[kStage];
    if (L > 1) {
        const f32x4* src = reinterpret_cast<const f32x4*>(a.wpack + a.bwd_off[L - 1]) + kHalf;
#pragma unroll
        for (int k = 0; k < kStage; ++k) {
            const int i = tid + 512 * k;
            if (i < kHalf) wstg[k] = src[i];
        }
    }
    constexpr int kVW = 14;           // H / 2 <= 56 hidden units over four k phases
    f32x4 vw[kVW];
    {
        const int cq = tid & 127, kg = tid >> 7;
        const f32x4* wq = reinterpret_cast<const f32x4*>(a.v0_w) + cq;
#pragma unroll
        for (int j = 0; j < kVW; ++j) {
            const int k = kg + 4 * j;
            vw[j] = (a.mode != 2 && cq < H && k < H / 2) ? wq[(size_t)k * H] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    const int r0 = a.gptr[gi], r1 = a.gptr[gi + 1];
    const int cnt = r1 - r0;
    if (cnt > kRows) { if (tid == 0) atomicOr(a.status, 2); return; }
    const int lrow = wave * 16 + r;
    const bool rvalid = lrow < cnt;
    const bool wactive = wave * 16 < cnt;
    const bool spare = cnt <= kRows / 2;            // workgroup-uniform: waves 4-7 own no rows (dma_share)
    const int grow = r0 + lrow;
    const int H2 = H / 2;
    const size_t slab = (size_t)a.n * HP;
    // Every global value the head-tail backward needs is requested here, before the CSR / weight staging, so that the
    // chain below waits for ONE memory round trip instead of one per barrier-separated step.
    const float dq_t = tid < cnt ? a.dq[r0 + tid] : 0.f;
    const float advr_t = tid < cnt ? a.adv_raw[r0 + tid] : 0.f;
    const float linw_t = (tid < 128 && tid < H) ? a.lin_w[tid] : 0.f;
    float vraw_g = 0.f, doutv_g = 0.f, z_t = 0.f, v1w_t = 0.f;
    int ax_t = -1, an_t = -1;
    if (a.mode != 2) {
        vraw_g = a.vraw[gi];
        if (a.mode != 0) doutv_g = a.d_out_v[gi];
        if (tid < H2) { z_t = a.z[(size_t)gi * H2 + tid]; v1w_t = a.v1_w[tid]; }
        if (tid < H) { ax_t = a.amax[(size_t)gi * H + tid]; an_t = a.amin[(size_t)gi * H + tid]; }
    }
    const float idg = rvalid ? a.invdeg[grow] : 0.f;     // (used from the first layer on: requested with everything else)
    f32x4 ytop[NT];      // y rows of the top layer: operand of the advantage-linear gradient and of the first ReLU mask
    {
        const f32x4* yr = reinterpret_cast<const f32x4*>(a.acts + slab * (L - 1) + (size_t)grow * HP) + g;
#pragma unroll
        for (int t = 0; t < NT; ++t) ytop[t] = rvalid ? yr[4 * t] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int e0 = a.rowptr_t[r0], ne = a.rowptr_t[r1] - e0;
    const bool csr_lds = load_csr<NT>(lds, a.rowptr_t, a.col_t, r0, cnt, e0, ne, a.status);
    float* s_max = reinterpret_cast<float*>(lds + LD::off_max);      // per-wave maxima (math 1)
    if (tid < 16) s_max[tid] = 0.f;
    if (tid < XS) dbuf[kRows * XS + tid] = 0.f;                      // the gather's filler row

    // stage the W_r part of the top layer into half B (the self half runs first; everything else arrives by LDS-DMA)
    if (L > 1) {
#pragma unroll
        for (int k = 0; k < kStage; ++k) { const int i = tid + 512 * k; if (i < kHalf) wbuf[kHalf + i] = wstg[k]; }
    }

    // ---- head tail backward; scratch aliases dbuf (not written before the first barrier A) ----
    float* sc = dbuf;
    float* s_w = sc;                  // [128]
    float* s_dp = sc + 128;           // [4*128]
    float* s_dz = sc + 640;           // [64]
    float* s_red = sc + 704;          // [8]
    float* s_dar = sc + 768;          // [128]
    int* s_ax = reinterpret_cast<int*>(sc + 896);    // [128] local row of the max
    int* s_an = reinterpret_cast<int*>(sc + 1024);
    float* s_lin = sc + 1152;         // [8][HP+1]
    float* s_part = sc + ((1152 + 8 * (HP + 1) + 3) & ~3);   // [3][HP] float4 partial sums of the value-MLP product (narrow widths: the row buffer is small)
    if (tid < 128) s_w[tid] = linw_t;
    float mean_dq = 0.f;
    const float inv_cnt = 1.f / (float)max(cnt, 1);
    if (a.mode != 2) {
        float ps = wsum64(dq_t);
        if (lane == 0) s_red[wave] = ps;
        if (tid < H) {
            s_ax[tid] = ax_t >= 0 ? ax_t - r0 : -1;
            s_an[tid] = an_t >= 0 ? an_t - r0 : -1;
        }
        __syncthreads();
        float sdq = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) sdq += s_red[w];
        mean_dq = sdq * inv_cnt;
        const float dV = a.mode == 0 ? sdq : doutv_g;
        const float dv = dV * sech2f(vraw_g);
        if (tid == 0) a.dvr[gi] = dv;
        if (tid < H2) {
            const float d = z_t > 0.f ? v1w_t * dv : 0.f;
            s_dz[tid] = d;
            a.dz[(size_t)gi * H2 + tid] = d;
        }
        __syncthreads();
        // d pooled = v0_w^T dz  ([H2] x [H2][4H]): 16-byte column groups x four k phases, every load independent
        {
            const int cq = tid & 127, kg = tid >> 7;
            f32x4 p4 = f32x4{0.f, 0.f, 0.f, 0.f};
            if (cq < H) {
#pragma unroll
                for (int j = 0; j < kVW; ++j) { const int k = kg + 4 * j; if (k < H2) p4 += vw[j] * s_dz[k]; }
            }
            if (kg > 0 && cq < H) reinterpret_cast<f32x4*>(s_part)[(kg - 1) * HP + cq] = p4;
            __syncthreads();
            if (kg == 0 && cq < H) {
                const f32x4* sp = reinterpret_cast<const f32x4*>(s_part) + cq;
                p4 += sp[0]; p4 += sp[HP]; p4 += sp[2 * HP];      // fixed order: deterministic
                // s_dp = [sum | max | min | mean][128]: each pooled block on its own 16-byte aligned row
#pragma unroll
                for (int j = 0; j < 4; ++j) { const int f = 4 * cq + j; s_dp[(f / H) * 128 + f % H] = p4[j]; }
            }
        }
    }
    if (tid < kRows) {
        float dar = 0.f;
        if (tid < cnt) {
            dar = (dq_t - mean_dq) * 2.f * sech2f(advr_t);
            a.dadv[r0 + tid] = dar;
        }
        s_dar[tid] = dar;
    }
    __syncthreads();
    const NbrRegs nbr = csr_lds ? load_nbrs<XS>(s_rp, s_col, lrow, rvalid, g)
                                : load_nbrs_global<XS>(a.rowptr_t, a.col_t, r0, cnt, e0, lrow, rvalid, g);

    // gradient w.r.t. the top layer's output, in the chained lane layout; advantage-linear partial alongside
    f32x4 gx[NT];
    {
        const float dar = s_dar[lrow];
        float lacc[NT * 4];
        typedef int i32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const f32x4 w = reinterpret_cast<const f32x4*>(s_w)[4 * t + g];
            const f32x4 yv = ytop[t];
            // this lane's four columns of the pooled gradients in 16-byte reads (as scalars: 168 ds_read_b32 per lane)
            const f32x4 d_sum = reinterpret_cast<const f32x4*>(s_dp)[4 * t + g];
            const f32x4 d_max = reinterpret_cast<const f32x4*>(s_dp + 128)[4 * t + g];
            const f32x4 d_min = reinterpret_cast<const f32x4*>(s_dp + 256)[4 * t + g];
            const f32x4 d_mean = reinterpret_cast<const f32x4*>(s_dp + 384)[4 * t + g];
            const i32x4 axv = reinterpret_cast<const i32x4*>(s_ax)[4 * t + g];
            const i32x4 anv = reinterpret_cast<const i32x4*>(s_an)[4 * t + g];
            f32x4 v;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int c = 16 * t + 4 * g + q4;
                float s = dar * w[q4];
                if (a.mode != 2 && c < H) {
                    s += d_sum[q4] + d_mean[q4] * inv_cnt;
                    if (axv[q4] == lrow) s += d_max[q4];
                    if (anv[q4] == lrow) s += d_min[q4];
                }
                v[q4] = (rvalid && c < H) ? s : 0.f;
                lacc[4 * t + q4] = dar * yv[q4];
            }
            gx[t] = v;
        }
        // d lin_w[c] partial = sum_rows dar*h[row][c]: reduce over the 16 rows of the wave, then over waves
        // (DPP row operations: the 16 rows of a wave are the 16 lanes of one DPP row; as __shfl_xor this butterfly was 116
        // ds_bpermute per wave and took 7.7 us of the prologue)
        float bacc = row16_sum(g == 0 ? dar : 0.f);
#pragma unroll
        for (int k = 0; k < NT * 4; ++k) lacc[k] = row16_sum(lacc[k]);
        if (r == 0) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) s_lin[wave * (HP + 1) + 16 * t + 4 * g + q4] = lacc[4 * t + q4];
            if (g == 0) s_lin[wave * (HP + 1) + HP] = bacc;
        }
    }
    __syncthreads();
    if (tid <= HP) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) s += s_lin[w * (HP + 1) + tid];
        a.lin_part[(size_t)gi * (HP + 1) + tid] = s;
    }
    __syncthreads();   // scratch consumed; dbuf may be overwritten from here on

    // ---- layer chain, in the forward kernel's shape ----
    //   dL/dy_{l-1} = [ T(G_l / deg) | G_l ] [W_l ; W_r]   (T = gather over the transposed CSR; linear, so the gather is
    //   moved in front of the contraction): per layer  gather from LDS -> K-half over the W_l part -> barrier ->
    //   K-half over the W_r part with G_l from registers -> mask by y_{l-1} -> publish G_{l-1} (global + LDS) -> barrier.
    // gx = dL/dy_l, yv = this lane's y_l chunks  ->  G_l = gx * [y_l > 0]; (l >= 1) G_l / deg goes to this lane's LDS row
    // for the neighbours' gathers.  The y rows are loaded by the caller a whole layer ahead into iteration-local
    // registers (a loop-carried prefetch made hipcc wait for the load in place).  store_G() then writes G_l for the
    // weight-gradient GEMM; it is a separate step so that the weight-half LDS writes can sit between the two (see the
    // forward kernel: a wait for staged loads placed after global stores also waits for the stores).
    const unsigned rowoff = (unsigned)grow * (HP * 4) + 16 * g;      // byte offset of this lane's slot inside a [n][HP] slab
    const unsigned lane16 = 16 * lane;
    const unsigned rowoff_v = rvalid ? rowoff : kOob;    // pad rows: stores dropped, loads return zeros
    // (no `if (rvalid)` region: a pad row's y loads return zeros, so the mask alone zeroes its gradient, and the tap store
    // drops out of range -- a divergent region around the mask cost a select or a copy per register at its merge)
    auto mask_rows = [&](const int l, const f32x4 (&yv)[NT]) {
        if (a.d_embeds && l == a.body_layers - 1) {
            const __amdgpu_buffer_rsrc_t de = slab_rsrc(a.d_embeds);
#pragma unroll
            for (int t = 0; t < NT; ++t) buf_store(gx[t], de, rowoff_v + 64 * t);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) gx[t][q4] = yv[t][q4] > 0.f ? gx[t][q4] : 0.f;
        }
        if (l > 0) {
            f32x4* dr = reinterpret_cast<f32x4*>(dbuf + lrow * XS) + g;
#pragma unroll
            for (int t = 0; t < NT; ++t) dr[4 * t] = gx[t] * idg;
            if constexpr (MATH == 1) {
                if (a.gmax) {
                    const float wm = rows_max16(row_max4(frag_absmax<NT>(gx, 0.f)));
                    if (lane == 0) s_max[wave] = wm;
                }
            }
        }
    };
    auto store_G = [&](const int l, const int t) {       // chunk t of this lane's row of G_l (held in gx)
        buf_store(gx[t], slab_rsrc(a.G + slab * l), rowoff_v + 64 * t);
    };
    QSTAMP(1, 0, 1);
    if (wactive) {
        mask_rows(L - 1, ytop);
#pragma unroll
        for (int t = 0; t < NT; ++t) store_G(L - 1, t);
    }
    lds_barrier();
    QSTAMP(1, 0, 2);
    // Per layer two phases in the forward kernel's shape (fillers between the MFMA groups):
    //   phase S: G_l (registers) x W_r part (half B); fillers = transposed LDS gather of G_l / deg, the deferred store of
    //            G_l, LDS-DMA of the W_l part into half A                                          -> barrier 1
    //   phase A: gathered rows x W_l part (half A); fillers = loads of y_{l-1}, LDS-DMA of layer l-1's W_r part into
    //            half B; then mask by y_{l-1}, G_{l-1} / deg -> LDS                                -> barrier 2
    constexpr int kGaps = Gaps<NT, MATH>::value;
    constexpr int kDma = (NT * NT + 7) / 8;
    constexpr int kFill = kDma + NT;
    constexpr int kTail = kFill > kGaps ? kGaps : kFill;
    const unsigned lds_w = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(lds + LD::off_w);
    for (int l = L - 1; l >= 1; --l) {
        QSTAMP(1, l, 0);
        const f32x4* wsrc = reinterpret_cast<const f32x4*>(a.wpack + a.bwd_off[l]);
        const bool more = l - 1 >= 1;
        const f32x4* nsrc = reinterpret_cast<const f32x4*>(a.wpack + a.bwd_off[more ? l - 1 : l]) + kHalf;
        if constexpr (MATH == 1) {
            if (a.gmax && tid == 0) {   // layer maximum of |G_l| over this graph -> global (order-independent)
                float mm = 0.f;
#pragma unroll
                for (int w8 = 0; w8 < 8; ++w8) mm = fmaxf(mm, s_max[w8]);
                atomicMax(a.gmax + l, __builtin_bit_cast(unsigned, mm));
            }
        }
        auto dmaS = [&](auto qq) {
            const int p = dma_share<NT>(wave, decltype(qq)::value, spare);
            if (p >= 0) dma_piece(wsrc + p * 64, lane16, lds_w + p * 1024);
            const int p2 = dma_share2<NT>(wave, decltype(qq)::value, spare);
            if (p2 >= 0) dma_piece(wsrc + p2 * 64, lane16, lds_w + p2 * 1024);
        };
        auto dmaA = [&](auto qq) {
            const int p = dma_share<NT>(wave, decltype(qq)::value, spare);
            if (more && p >= 0) dma_piece(nsrc + p * 64, lane16, lds_w + (kHalf + p * 64) * 16);
            const int p2 = dma_share2<NT>(wave, decltype(qq)::value, spare);
            if (more && p2 >= 0) dma_piece(nsrc + p2 * 64, lane16, lds_w + (kHalf + p2 * 64) * 16);
        };
        if (!wactive) {      // a wave without rows only moves its weight pieces (own straight path, as in the forward kernel)
            static_for<0, kDma>(dmaS);
            wait_vmem();
            lds_barrier();
            static_for<0, kDma>(dmaA);
            wait_vmem();
            lds_barrier();
            continue;
        }
        f32x4 acc[NT], ag[NT];
        if constexpr (MATH == 1) {
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        float rs = 1.f, rinv = 1.f, mx = 0.f;
        f32x4 tb[GatherLand<NT, MATH>::value][NT];
        if constexpr (MATH == 1) { mx = row_max4(frag_absmax<NT>(gx, 0.f)); row_scale(mx, rs, rinv); }
        // ---- phase S ----
        const __amdgpu_buffer_rsrc_t gcur = slab_rsrc(a.G + slab * l);
        const unsigned gcur_off = l < L - 1 ? rowoff_v : kOob;      // the top layer's G was stored ahead of the loop
        auto fillS = [&](auto qq) {
            constexpr int Q = decltype(qq)::value;
            gather_gap<NT, Q, kGaps, MATH>(dbuf, nbr, ag, tb);
            if constexpr (Q < kDma) dmaS(qq);
            else if constexpr (Q < kDma + NT) buf_store(gx[Q - kDma], gcur, gcur_off + 64 * (Q - kDma));
        };
        contract_half_fill<NT, MATH, MATH == 0>(wbuf + kHalf, lane, gx, acc, rs, fillS);
        static_for<kTail, kFill>(fillS);
        if (nbr.wlong) {      // (wave-uniform; pad rows: eb == ee)
            if (csr_lds) gather_lds<NT, XS>(dbuf, s_col, nbr.eb, nbr.ee, g, ag);
            else gather_global_tail<NT, XS>(dbuf, a.col_t, r0, cnt, e0 + nbr.eb, e0 + nbr.ee, g, ag);
        }
        QSTAMP(1, l, 1);
        wait_vmem();
        QSTAMP(1, l, 3);
        lds_barrier();     // barrier 1: half A = W_l part; every gather of this layer is done (dbuf free); half B free
        QSTAMP(1, l, 4);
        // ---- phase A ----
        float rsa = 1.f;
        if constexpr (MATH == 1) {   // own power-of-two row scale for the gathered rows, exact carry of the self half's sums
            float ma = row_max4(frag_absmax<NT>(ag, 0.f));
            ma = fmaxf(ma, mx * 0x1p-40f);
            float rinva;
            row_scale(ma, rsa, rinva);
            const float carry = rsa * rinv;
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] *= carry;
            rinv = rinva * (reinterpret_cast<const float*>(a.wpack + a.bias_off[l]) + HP)[1];
        }
        const __amdgpu_buffer_rsrc_t yr = slab_rsrc(a.acts + slab * (l - 1));
        f32x4 yl[NT];        // y_{l-1} rows for this iteration's closing mask
        auto fillA = [&](auto qq) {
            constexpr int Q = decltype(qq)::value;
            if constexpr (Q < NT) {
                load_guard<MATH>();
                yl[Q] = buf_load(yr, rowoff_v + 64 * Q);
            } else if constexpr (Q < NT + kDma) {
                dmaA(std::integral_constant<int, Q - NT>{});
            }
        };
        contract_half_fill<NT, MATH>(wbuf, lane, ag, acc, rsa, fillA);
        static_for<kTail, kFill>(fillA);
        QSTAMP(1, l, 5);
#pragma unroll
        for (int t = 0; t < NT; ++t) gx[t] = MATH == 1 ? acc[t] * rinv : acc[t];
        mask_rows(l - 1, yl);
        wait_vmem();
        QSTAMP(1, l, 6);
        lds_barrier();     // barrier 2: G_{l-1} rows + half B visible; half A free
        QSTAMP(1, l, 7);
    }
    if (wactive && L > 1) {
#pragma unroll
        for (int t = 0; t < NT; ++t) store_G(0, t);
    }
    // ---- raw first layer: this graph's share of dW_0 = G_0^T [agg0 | x0 | 1], reduced over the graphs afterwards ----
    if (a.first_part) {
        __syncthreads();             // every wave's last gather is done: dbuf and the weight halves are free
        // F = [agg0(8) | x0(8) | 1 | 0...] per row, 48-float rows (stride == 16 mod 32: conflict-free fragment reads)
        float* s_f = reinterpret_cast<float*>(lds + LD::off_scr_bwd0);
        {
            f32x4* dr = reinterpret_cast<f32x4*>(dbuf + lrow * XS) + g;
#pragma unroll
            for (int t = 0; t < NT; ++t) dr[4 * t] = rvalid ? gx[t] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (tid < kRows) {
            f32x4* fr = reinterpret_cast<f32x4*>(s_f + tid * 48);
            f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f}, a1 = a0, x0 = a0, x1 = a0, one = a0;
            if (tid < cnt) {
                const f32x4* ar = reinterpret_cast<const f32x4*>(a.agg0 + (size_t)(r0 + tid) * kSmallCin);
                a0 = ar[0]; a1 = ar[1];
                const float* xr = a.x + (size_t)(r0 + tid) * a.x_stride;
#pragma unroll
                for (int q = 0; q < 4; ++q) { x0[q] = q < a.c_in ? xr[q] : 0.f; x1[q] = 4 + q < a.c_in ? xr[4 + q] : 0.f; }
                one[0] = 1.f;
            }
            fr[0] = a0; fr[1] = a1; fr[2] = x0; fr[3] = x1; fr[4] = one;
            fr[5] = f32x4{0.f, 0.f, 0.f, 0.f}; fr[6] = fr[5]; fr[7] = fr[5];
        }
        __syncthreads();
        if (wave < NT) {
            // dW_0^T tile [c (2 x 16)][o = 16 wave ..] = F^T G_0 over the 128 rows: exact fp32 MFMA, two interleaved chains
            f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll 8
            for (int k = 0; k < kRows / 4; ++k) {
                const int row = 4 * k + g;
                const float bv = dbuf[row * XS + 16 * wave + r];
                acc0 = mfma16x16x4(s_f[row * 48 + r], bv, acc0);
                acc1 = mfma16x16x4(s_f[row * 48 + 16 + r], bv, acc1);
            }
            float* out = a.first_part + (size_t)gi * 17 * HP + 16 * wave + r;
#pragma unroll
            for (int j = 0; j < 4; ++j) out[(size_t)(4 * g + j) * HP] = acc0[j];
            if (g == 0) out[(size_t)16 * HP] = acc1[0];
        }
    }
    QSTAMP(1, 0, 3);
}

template <int NT, int MATH>
static int launch_qfwd_m(const QFwdArgs& a, hipStream_t st) {
    static bool once = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&qnet_fwd_kernel<NT, MATH>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, QLds<NT>::total);
        return true;
    }();
    (void)once;
    qnet_fwd_kernel<NT, MATH><<<a.b, 512, QLds<NT>::total, st>>>(a);
    return HEXGNN_OK;
}
template <int NT, int MATH>
static int launch_qbwd_m(const QBwdArgs& a, hipStream_t st) {
    static bool once = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&qnet_bwd_kernel<NT, MATH>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, QLds<NT>::total);
        return true;
    }();
    (void)once;
    qnet_bwd_kernel<NT, MATH><<<a.b, 512, QLds<NT>::total, st>>>(a);
    return HEXGNN_OK;
}

// one translation unit per math mode (co-compiled template variants perturb each other's register allocation)
int launch_qfwd_math(int nt, int math, const QFwdArgs& a, hipStream_t st);
int launch_qbwd_math(int nt, int math, const QBwdArgs& a, hipStream_t st);
int launch_qfwd_split(int nt, const QFwdArgs& a, hipStream_t st);
int launch_qbwd_split(int nt, const QBwdArgs& a, hipStream_t st);

}  // namespace hexgnn
