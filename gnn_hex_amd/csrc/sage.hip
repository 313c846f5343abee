// GraphSAGE stack for gfx950: fused (CSR mean-gather -> fp32 MFMA [agg|x]*[Wl;Wr]^T -> bias -> ReLU) per
// layer, its data-gradient twin, and a batched weight-gradient GEMM.
//
// Reference semantics: CachifiedGNN.forward (GN0/models.py:261-294) over pyg SAGEConv
// (GN0/torch_script_models.py:52-73):  y_i = W_l mean_{j in N(i)} x_j + b_l + W_r x_i ; ReLU every layer.
//
// Data layout (HBM): every node-feature matrix is [n][HP] fp32, HP = 16*NT (pads zero).  A workgroup is
// 8 waves; wave w owns the 16-row block 8*blockIdx+w.  The whole packed weight matrix of the layer
// (2*NT*NT KiB) is staged once per workgroup into LDS in MFMA-fragment order, so every B fragment is one
// conflict-free ds_read_b128.  The A operand never touches LDS: lane (r = l&15, g = l>>4) gathers, for
// its row r, the feature chunks {16c+4g .. 16c+4g+3} of every in-neighbour straight into registers; these
// four floats are the k-slices of four consecutive v_mfma_f32_16x16x4_f32 (the K order is permuted
// consistently on the packed-weight side).  fp32 in / fp32 accumulate: exact fmaf chains, deterministic.
#include <atomic>
#include <mutex>
#include <type_traits>
#include <utility>
#include "hexgnn_internal.h"
#include "hexgnn_pack.h"
#include "hexgnn_memops.h"
#include "sage_dw_kernel.h"

namespace hexgnn {

int make_plan(int n, int c_in, int hidden, int L, StackPlan* p) {
    const int hp = padded_width(hidden);
    if (hp < 0 || L < 1 || L > kMaxLayers) return HEXGNN_EUNSUPPORTED;
    if (c_in != hidden && (c_in < 1 || c_in > kSmallCin)) return HEXGNN_EUNSUPPORTED;
    // rows are addressed through raw-buffer resources with 32-bit byte offsets and num_records 2^31 - 1: a slab of n rows must
    // stay below 2 GiB (4.7 M nodes at hidden 112: ~150x the largest BASELINE batch), beyond it loads would return zeros
    if (n > 0 && (size_t)n * hp * sizeof(float) > 0x7fffffffull) return HEXGNN_EUNSUPPORTED;
    p->hp = hp; p->nt = hp / 16; p->L = L; p->c_in = c_in;
    p->small_first = (c_in != hidden);
    size_t off = 0, soff = 0;
    const size_t pack = (size_t)2 * p->nt * p->nt * 64 * sizeof(f32x4);
    for (int l = 0; l < L; ++l) {
        if (l == 0 && p->small_first) {
            p->fwd_off[l] = off; off += sizeof(float) * (size_t)hp * kSmallCin * 2;  // [HP][8] Wl, [HP][8] Wr
            p->bwd_off[l] = 0;
            p->agg_off[l] = soff; soff += align_up(sizeof(float) * (size_t)n * kSmallCin, 256);
        } else {
            p->fwd_off[l] = off; off += pack;
            p->bwd_off[l] = off; off += pack;
            p->agg_off[l] = soff; soff += align_up(sizeof(float) * (size_t)n * hp, 256);
        }
        p->bias_off[l] = off; off += align_up(sizeof(float) * (hp + 2), 256);   // bias[hp], then {w scale, 1/scale} (math 1)
    }
    p->flag_off = off; off += sizeof(unsigned) * 2 * kStackFlagWords;
    p->pack_bytes = off;
    p->saved_bytes = soff;
    return HEXGNN_OK;
}

// ---- weight packing (one launch per stack call; grid.y = layer): body in hexgnn_pack.h -------------------------------------
__global__ void sage_pack_kernel(PackArgs a, char* __restrict__ wpack) {
    sage_pack_body(a, wpack, blockIdx.x, blockIdx.y, gridDim.x);
}

// ---- split-precision packing for the fused kernels (math mode 1, "f16x3"): every fp32 weight w of a layer is scaled
//      by the layer's power of two s_W (max |w| * s_W in [2^14, 2^15)) and stored as two fp16 planes hi = f16(w s_W),
//      lo = f16(w s_W - hi): 22 significand bits.  The contraction W*X ~= Whi*Xhi + Whi*Xlo + Wlo*Xhi runs on the f16
//      MFMA pipe with fp32 accumulation (rows get their own power-of-two scale in the kernel; both are undone exactly in
//      the epilogue).  Product error ~3*2^-22 relative to max|w| max|x|; measured parity in tests/test_gpu_model.py.
//      Layout per K-half (NT*NT KiB, identical size to the fp32 pack): units u = chunk pairs (2p, 2p+1) [+ one odd
//      chunk]; unit of s chunks at byte (first_chunk*NT) KiB; tile t at + t*s KiB; plane hi at +0, lo at + s*512 B;
//      lane l at + l*8*s B holding the k-slots (kq = l>>4): j < 4 -> feature 16*c0 + 4*kq + j, j >= 4 -> 16*(c0+1) + 4*kq + j-4.
struct Pack16Args {
    LayerPtrs p;
    size_t fwd_off[kMaxLayers], bwd_off[kMaxLayers], bias_off[kMaxLayers];
    int nt, L, hidden, first_hidden, hp;
    unsigned* zero_maxima;    // 2*kMaxLayers words cleared by the scale kernel (per-layer activation / gradient maxima), or null
};
__device__ __forceinline__ unsigned short f16_bits(float v) { _Float16 b = (_Float16)v; return __builtin_bit_cast(unsigned short, b); }
__device__ __forceinline__ float f16_to_f32(unsigned short u) { return (float)__builtin_bit_cast(_Float16, u); }

// per hidden layer: s_W = 2^(14 - floor(log2 max|w|)) over W_l and W_r, stored with its inverse after the padded bias
__global__ __launch_bounds__(1024) void sage_wscale_kernel(Pack16Args a, char* __restrict__ wpack) {
    const int l = a.first_hidden + blockIdx.x;
    const int H = a.hidden;
    const float* wl = a.p.wl[l];
    const float* wr = a.p.wr[l];
    if (blockIdx.x == 0 && a.zero_maxima && threadIdx.x < 2 * kMaxLayers) a.zero_maxima[threadIdx.x] = 0u;
    float m = 0.f;
    for (int i = threadIdx.x; i < H * H; i += 1024) m = fmaxf(m, fmaxf(fabsf(wl[i]), fabsf(wr[i])));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __shared__ float sm[16];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < 16; ++w) m = fmaxf(m, sm[w]);
        const unsigned e = __builtin_bit_cast(unsigned, m) >> 23;
        const bool ok = e >= 64u && e <= 190u;
        float* out = reinterpret_cast<float*>(wpack + a.bias_off[l]) + a.hp;
        out[0] = ok ? __builtin_bit_cast(float, (268u - e) << 23) : 1.f;
        out[1] = ok ? __builtin_bit_cast(float, (e - 14u) << 23) : 1.f;
    }
}

__global__ void sage_pack16_kernel(Pack16Args a, char* __restrict__ wpack) {
    const int l = a.first_hidden + blockIdx.y;
    const int nt = a.nt, H = a.hidden;
    const float* wl = a.p.wl[l];
    const float* wr = a.p.wr[l];
    const float wscale = reinterpret_cast<const float*>(wpack + a.bias_off[l])[a.hp];
    // one thread per (direction, half, chunk c, tile t, lane, j<4): 2*2*nt*nt*64*4 elements
    const int per_dir = 2 * nt * nt * 256;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= 2 * per_dir) return;
    const int dir = tid / per_dir;              // 0 forward, 1 backward
    int rem = tid % per_dir;
    const int half = rem / (nt * nt * 256); rem %= nt * nt * 256;
    const int c = rem / (nt * 256); rem %= nt * 256;
    const int t = rem / 256; rem %= 256;
    const int lane = rem >> 2, j = rem & 3;
    const int kq = lane >> 4, m = lane & 15;
    const float* w = half == 0 ? wl : wr;
    float v;
    if (dir == 0) {   // forward: k = input feature 16c+4kq+j, output o = 16t+m
        const int k = 16 * c + 4 * kq + j, o = 16 * t + m;
        v = (k < H && o < H) ? w[o * H + k] : 0.f;
    } else {          // backward: k = output o = 16c+4kq+j, produces input feature i = 16t+m
        const int o = 16 * c + 4 * kq + j, i = 16 * t + m;
        v = (o < H && i < H) ? w[o * H + i] : 0.f;
    }
    v *= wscale;
    const unsigned short hi = f16_bits(v);
    const unsigned short lo = f16_bits(v - f16_to_f32(hi));
    const bool paired = (c | 1) < nt;           // chunk belongs to a full pair
    const int c0 = c & ~1;
    const int s = paired ? 2 : 1;
    const int unit_first = paired ? c0 : c;
    char* base = wpack + (dir == 0 ? a.fwd_off[l] : a.bwd_off[l]) + (size_t)half * nt * nt * 1024
               + (size_t)unit_first * nt * 1024 + (size_t)t * s * 1024;
    const int jj = paired ? (c - c0) * 4 + j : j;
    unsigned short* ph = reinterpret_cast<unsigned short*>(base + lane * 8 * s) + jj;
    unsigned short* pl = reinterpret_cast<unsigned short*>(base + s * 512 + lane * 8 * s) + jj;
    *ph = hi;
    *pl = lo;
}

// ---- first layer, raw features (c_in <= 8): VALU, HBM-bound ----------------------------------------
// 32 rows per 256-thread workgroup.  Saves the aggregated raw features [n][8] for the backward pass.
__global__ __launch_bounds__(256) void sage_first_fwd_kernel(
    int n, int c_in, int hp, const int* __restrict__ rowptr, const int* __restrict__ col,
    const float* __restrict__ invdeg, const float* __restrict__ x, int x_stride,
    const float* __restrict__ w0 /*[hp][8] Wl then [hp][8] Wr*/, const float* __restrict__ bias,
    float* __restrict__ y, float* __restrict__ agg_out /*[n][8] or null*/, int relu) {
    __shared__ float sA[32][kSmallCin], sX[32][kSmallCin];
    __shared__ float sW[2 * 128 * kSmallCin + 128];
    const int tid = threadIdx.x;
    const int r0 = blockIdx.x * 32;
    for (int i = tid; i < 2 * hp * kSmallCin; i += 256) sW[i] = w0[i];
    for (int i = tid; i < hp; i += 256) sW[2 * 128 * kSmallCin + i] = bias[i];
    {
        // eight lanes per row: lane k takes neighbours k, k + 8, ... (one thread per row walked the CSR serially: 16 us for a
        // layer of 2 x 110 FMAs per node); the partial sums meet in a fixed xor tree over the eight lanes
        const int rr = tid >> 3, k = tid & 7;
        const int row = r0 + rr;
        float a[kSmallCin], s[kSmallCin];
#pragma unroll
        for (int q = 0; q < kSmallCin; ++q) { a[q] = 0.f; s[q] = 0.f; }
        if (row < n) {
            const int e1 = rowptr[row + 1];
            for (int e = rowptr[row] + k; e < e1; e += 8) {
                const float* xr = x + (size_t)col[e] * x_stride;
#pragma unroll
                for (int q = 0; q < kSmallCin; ++q) if (q < c_in) a[q] += xr[q];
            }
        }
#pragma unroll
        for (int q = 0; q < kSmallCin; ++q) {
            a[q] += __shfl_xor(a[q], 1);
            a[q] += __shfl_xor(a[q], 2);
            a[q] += __shfl_xor(a[q], 4);
        }
        if (k == 0) {
            if (row < n) {
                const float sc = invdeg[row];
                const float* xs = x + (size_t)row * x_stride;
#pragma unroll
                for (int q = 0; q < kSmallCin; ++q) { a[q] *= sc; if (q < c_in) s[q] = xs[q]; }
                if (agg_out) {
#pragma unroll
                    for (int q = 0; q < kSmallCin; ++q) agg_out[(size_t)row * kSmallCin + q] = a[q];
                }
            }
#pragma unroll
            for (int q = 0; q < kSmallCin; ++q) { sA[rr][q] = a[q]; sX[rr][q] = s[q]; }
        }
    }
    __syncthreads();
    const float* sWl = sW;
    const float* sWr = sW + hp * kSmallCin;
    const float* sB = sW + 2 * 128 * kSmallCin;
    for (int idx = tid; idx < 32 * hp; idx += 256) {
        const int r = idx / hp, c = idx % hp;
        const int row = r0 + r;
        if (row >= n) break;
        float v = sB[c];
#pragma unroll
        for (int q = 0; q < kSmallCin; ++q) v += sWl[c * kSmallCin + q] * sA[r][q] + sWr[c * kSmallCin + q] * sX[r][q];
        y[(size_t)row * hp + c] = (v > 0.f || !relu) ? v : 0.f;
    }
}


// acc[c] += rows[j][chunk c] over the CSR row [e0,e1): two neighbours per iteration, all 2*NT 16-byte loads issued
// before the first add (the column ids of the next pair are fetched ahead); ascending neighbour order is kept.
template <int NT>
__device__ __forceinline__ void gather_rows_global(const float* __restrict__ rows, const int* __restrict__ col, int e0,
                                                   int e1, int g, f32x4 (&acc)[NT]) {
    constexpr int HP = 16 * NT;
    int e = e0;
    while (e + 1 < e1) {
        const int j0 = col[e], j1 = col[e + 1];
        e += 2;
        const f32x4* x0 = reinterpret_cast<const f32x4*>(rows + (size_t)j0 * HP) + g;
        const f32x4* x1 = reinterpret_cast<const f32x4*>(rows + (size_t)j1 * HP) + g;
        f32x4 t0[NT], t1[NT];
#pragma unroll
        for (int c = 0; c < NT; ++c) t0[c] = x0[4 * c];
#pragma unroll
        for (int c = 0; c < NT; ++c) t1[c] = x1[4 * c];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < NT; ++c) acc[c] += t0[c];
#pragma unroll
        for (int c = 0; c < NT; ++c) acc[c] += t1[c];
    }
    if (e < e1) {
        const f32x4* x0 = reinterpret_cast<const f32x4*>(rows + (size_t)col[e] * HP) + g;
        f32x4 t0[NT];
#pragma unroll
        for (int c = 0; c < NT; ++c) t0[c] = x0[4 * c];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < NT; ++c) acc[c] += t0[c];
    }
}

template <int B, int E, typename F>
__device__ __forceinline__ void static_for_(F&& f) {
    if constexpr (B < E) { f(std::integral_constant<int, B>{}); static_for_<B + 1, E>(f); }
}

#ifdef HEXGNN_STAMPS
// profiling builds only (make STAMPS=1, tools/layer_stamps.py): lane 0 of every wave of one mid-grid workgroup records
// s_memtime at fixed points of the layer-major kernels (the last launch of each kind wins)
__device__ unsigned long long g_lstamps[2][8][8];
#define LSTAMP(k, p) do { if (blockIdx.x == gridDim.x / 2 && (threadIdx.x & 63) == 0) g_lstamps[k][p][threadIdx.x >> 6] = __builtin_amdgcn_s_memtime(); } while (0)
__device__ unsigned long long g_pstamps[2][16][8];      // one-launch stack kernels: layer 8 of the mid-grid workgroup (or of
__device__ int g_stamp_block = -1;                      // the one chosen with hexgnn_debug_stamp_block)
#define PSTAMP(k, p) do { if (it == 8 && (int)blockIdx.x == (g_stamp_block < 0 ? (int)gridDim.x / 2 : g_stamp_block) && (threadIdx.x & 63) == 0) g_pstamps[k][p][threadIdx.x >> 6] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define LSTAMP(k, p) do {} while (0)
#define PSTAMP(k, p) do {} while (0)
#endif

// The rest of a row with more than kEll neighbours (the terminals of a board; after dead / captured removal late in a game many
// rows), summed from global memory in CSR order.  Four neighbour ids (and, backward, their 1 / deg) are requested at once and
// DEPTH neighbour rows are in flight (the one-launch kernels sit at the register ceiling: one): per four neighbours 1 + 4 / DEPTH
// round trips instead of eight (the one-at-a-time loop made random-playout MIX batches 60 % slower than start positions: 413
// against 255 us per launch).  Same order of additions
// as the plain loop: bit-identical sums.  COH: agent-scope (sc1) row loads -- rows written by other workgroups of this launch.
#ifndef HEXGNN_LR_IDS
#define HEXGNN_LR_IDS 2
#endif
template <int NT, bool BWD, bool COH, int DEPTH, int IDS>
__device__ __forceinline__ void long_row_tail(const int* __restrict__ col, __amdgpu_buffer_rsrc_t ir /* 1 / deg (backward) */,
                                              __amdgpu_buffer_rsrc_t xr, int e_lo, int e_hi, int g, f32x4 (&ag)[NT]) {
    constexpr unsigned kRowB = 16u * NT * 4u;
    constexpr int kAux = COH ? 16 : 0;
    const __amdgpu_buffer_rsrc_t cr = slab_rsrc(col);
    for (int e = e_lo; e < e_hi; e += IDS) {
        int j[IDS];
        float sj[IDS];
#pragma unroll
        for (int q = 0; q < IDS; ++q)
            j[q] = __builtin_amdgcn_raw_buffer_load_b32(cr, e + q < e_hi ? (unsigned)(e + q) * 4u : kOob, 0, 0);
#pragma unroll
        for (int q = 0; q < IDS; ++q) {
            sj[q] = 1.f;
            if constexpr (BWD)
                sj[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ir, e + q < e_hi ? (unsigned)j[q] * 4u : kOob, 0, 0));
        }
#pragma unroll
        for (int h = 0; h < IDS; h += DEPTH) {
            if (e + h >= e_hi) break;             // (no lane of the wave left with a neighbour in this group: nothing is issued)
            f32x4 rr[DEPTH][NT];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const unsigned o = e + h + d < e_hi ? (unsigned)j[h + d] * kRowB + 16u * g : kOob;
#pragma unroll
                for (int c = 0; c < NT; ++c)
                    rr[d][c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, o + 64 * c, 0, kAux));
            }
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                if (e + h + d < e_hi) {
#pragma unroll
                    for (int c = 0; c < NT; ++c) {
                        if constexpr (BWD) ag[c] += rr[d][c] * sj[h + d];
                        else ag[c] += rr[d][c];
                    }
                }
            }
        }
    }
}

// ---- hidden layer, forward and backward (data) -------------------------------------------------------------------
//   forward :  y_i   = act( mean_{j in N(i)} x_j W_l^T + x_i W_r^T + b )                       (GN0/torch_script_models.py:52-73)
//   backward:  dY_i  = ( sum_{j in T(i)} G_j / deg_j ) W_l + G_i W_r,   G' = dY * [y' > 0]       (its autograd transpose, with
//              the gather moved in front of the contraction: it is linear, and the kernel then has the forward's shape)
// One 128-row block per 512-thread workgroup, wave w = rows 16w..16w+15, lane (r, g) = row r, 16-byte column slots g, g+4, ...
// Timeline of a launch (tools/layer_stamps.py; the straight-line version spent 3.7 us staging weights, 9.5 us in the gather's
// dependent chain rowptr -> column ids -> neighbour rows and 13.5 us in MFMAs, one after the other: 29 us):
//   1. both weight parts -> LDS by LDS-DMA (no registers, nothing waits), self rows and CSR row bounds requested alongside:
//      ONE memory round trip, then the barrier;
//   2. self half (rows x W_r part) on the matrix pipe while the gather runs underneath it: the column ids of the row's
//      first sixteen neighbours, then the neighbour rows as single 16-byte-slot loads spread evenly over the gaps between
//      MFMA groups (raw buffer loads; a missing neighbour is an out-of-range offset = zeros, so every lane issues the same
//      instructions), added in ascending neighbour order a few gaps later;
//   3. rows with more than sixteen neighbours finish their sum from the CSR, then the aggregate half, epilogue.
constexpr int kEll = 16;          // neighbour slots handled inside the self half (longer rows finish from the CSR)

// MFMA order of a K-half (NT >= 3): the NT*NT units (chunk c, tile t) of four dependent MFMAs each are issued in GROUPS of
// three units round robin -- u0.j0 u1.j0 u2.j0 u0.j1 ... u2.j3 -- so that two MFMAs on the same accumulator are always at
// least three slots (>= 96 cycles of matrix-pipe time) apart, more than the instruction's 40-cycle dependent latency (the
// last group takes the NT*NT mod 3 = 1 leftover unit as a fourth member; across group boundaries the distance is >= 3 too:
// units three apart never share a tile for NT > 3, and for NT = 3 the same tile returns exactly three slots later).  A
// unit's four MFMAs issued back to back (the earlier order) each waited 8 cycles inside the pipe for their srcC, and an MFMA
// that waits there reads srcC late: when hipcc let its accumulator move (vdst != srcC) and handed the dead srcC registers to
// the next load, nothing interlocked the load's return against the pending read (the width-24 bug of
// the fused kernels; tools/scan_mfma_war.py).  With every dependent pair >= 3 slots apart an MFMA's srcC is complete when
// it issues.  Per accumulator the k order is unchanged (same fmaf chains, same bits).  Three weight fragments are live (four
// in the last group) instead of one; a gap (filler slot) follows every four MFMAs, NT*NT gaps per half as before.
template <int NT> struct MfmaSeq {
    static constexpr int U = NT * NT;
    static constexpr int kGroups = U / 3;                      // NT >= 3
    static constexpr int group_size(int g) { return g + 1 < kGroups ? 3 : U - 3 * (kGroups - 1); }   // 3, or 4 at the end
    static constexpr int kGaps = U;
};

// Gather schedule of the self half, neighbours fetched from GLOBAL memory (hidden 113..128, where the row copy below does
// not fit beside the weights): the W * NT 16-byte neighbour loads of a lane are issued kP per gap (a gap = the slot behind
// four MFMAs; a burst of loads instead would hold the wave -- and with it its MFMAs -- in the CU's 64 B/clk vector-memory
// issue path), neighbour k lands in buffer k % kWin and is added kD gaps after its last load, before the first load of
// neighbour k + kWin in the same buffer.  Gaps past the last MFMA group run behind the loop.
template <int NT> struct GatherSched {
    static constexpr int W = kEll;
    static constexpr int G = NT * NT;
    static constexpr int kWin = NT >= 8 ? 3 : (NT >= 4 ? 4 : 8);     // landing buffers (three at hidden 113-128: four spill)
    static constexpr int kSpan = (7 * G) / 8 > 0 ? (7 * G) / 8 : 1;
    static constexpr int kPspan = (W * NT + kSpan - 1) / kSpan;
    static constexpr int kPmax = (kWin * NT - NT + 1) / 2;           // keeps kD >= 1: an add never shares a gap with its loads
    static constexpr int kP = kPspan < kPmax ? kPspan : kPmax;
    static constexpr int kD = (kWin * NT - (NT - 1)) / kP - 1;
    static constexpr int load_gap(int k, int c) { return (k * NT + c) / kP; }
    static constexpr int add_gap(int k) { return load_gap(k, NT - 1) + kD; }
    static constexpr int kGaps = add_gap(W - 1) + 1 > G ? add_gap(W - 1) + 1 : G;
    static constexpr bool ok() {
        if (kP < 1 || kD < 1) return false;
        for (int k = 0; k + kWin < W; ++k)
            if (add_gap(k) > load_gap(k + kWin, 0)) return false;      // (adds run before the loads of a gap)
        return true;
    }
    static_assert(ok(), "a landing buffer would be reloaded before it is consumed");
};

// Up to hidden 112 the block's OWN 128 rows are kept in LDS beside the weights (98 KB + 129 x 464 B = 157 KB at NT = 7; row
// 128 is all zero): a board graph's neighbours sit within a few dozen rows of the node, so most of a block's neighbour reads
// stay inside the block and become LDS reads; only rows near a block boundary (and the two terminal rows of a graph cut by
// it) still fetch from global memory.  Round 2's kernels read EVERY neighbour row through L1/L2: 16 slots x NT loads per
// lane, the self half took 21-30 k ticks against 12.5 k of MFMAs (profiles/r02/layer_stamps_MIX.txt).
//   slot k: LDS read in gap k * stride (out-of-block lanes read the zero row), added one gap later;
//           global load in the same gap for the lanes that need it -- skipped wave-uniformly (a 16-bit mask of ballots) when
//           no lane of the wave does -- into a ring of two landing buffers, added kGd gaps later (the ring of four of the
//           all-global schedule would not fit the registers beside the second offset table).
template <int NT> struct RowsLds {
    static constexpr bool on = NT >= 4 && NT <= 7;       // (narrower: hipcc spills the second offset table; wider: no LDS left)
    static constexpr int XS = 16 * NT + 4;                   // floats per row: an odd number of 16-byte slots
    static constexpr int bytes = on ? 129 * XS * 4 : 0;
    static_assert(!on || 129 * XS * 4 <= 65536, "row offsets are kept as u16");
};
template <int NT> struct GatherLds {
    static constexpr int G = NT * NT;
    static constexpr int stride = (G - 3) / kEll > 0 ? (G - 3) / kEll : 1;
    static constexpr int kGd = 2 * stride < 4 ? 2 * stride : 4;          // adds run before the loads of a gap: a ring of TWO is safe
    static constexpr int rd_gap(int k) { return k * stride; }
    static constexpr int add_gap(int k) { return k * stride + 1; }
    static constexpr int gadd_gap(int k) { return k * stride + kGd; }
    static constexpr int kGaps = gadd_gap(kEll - 1) + 1 > G ? gadd_gap(kEll - 1) + 1 : G;
};

template <int NT, bool BWD>
__device__ __forceinline__ void sage_layer_body(
    int n, const int* __restrict__ rowptr, const int* __restrict__ col,
    const float* __restrict__ invdeg, const float* __restrict__ x, const f32x4* __restrict__ wpack,
    const float* __restrict__ bias, const float* __restrict__ ymask, float* __restrict__ out,
    float* __restrict__ agg_out, int relu, f32x4* wlds) {
    constexpr int HP = 16 * NT;
    constexpr int K = BWD ? 1 : 0;     // stamp set (profiling builds)
    (void)K;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    LSTAMP(K, 0);
    {
        const unsigned lds_w = (unsigned)(size_t)(__attribute__((address_space(3))) char*)wlds;
        for (int p = wave; p < 2 * NT * NT; p += 8) dma_piece(wpack + p * 64, 16 * lane, lds_w + p * 1024);
    }
    const int row0 = (blockIdx.x * 8 + wave) * 16;
    const int r = lane & 15, g = lane >> 4;
    const int row = row0 + r;
    const bool valid = row < n;
    f32x4 xs[NT], ag[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c) { xs[c] = f32x4{0.f, 0.f, 0.f, 0.f}; ag[c] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    int e0 = 0, e1 = 0;
    int nid[kEll];
    float sc = 0.f;
    if (valid) {
        const f32x4* xr = reinterpret_cast<const f32x4*>(x + (size_t)row * HP) + g;
#pragma unroll
        for (int c = 0; c < NT; ++c) xs[c] = xr[4 * c];
        e0 = rowptr[row];
        e1 = rowptr[row + 1];
        if constexpr (!BWD) sc = invdeg[row];
    }
    using RL = RowsLds<NT>;
    float* rowsl = reinterpret_cast<float*>(wlds + 2 * NT * NT * 64);      // [129][XS] behind the weights (NT <= 7)
    if constexpr (RL::on) {
        f32x4* mine = reinterpret_cast<f32x4*>(rowsl + (wave * 16 + r) * RL::XS) + g;
#pragma unroll
        for (int c = 0; c < NT; ++c) mine[4 * c] = xs[c];                   // (rows past n are zeros)
        if (tid < RL::XS / 4) reinterpret_cast<f32x4*>(rowsl + 128 * RL::XS)[tid] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    wait_vmem();
    LSTAMP(K, 1);
    __syncthreads();
    LSTAMP(K, 3);

    const __amdgpu_buffer_rsrc_t xr_ = slab_rsrc(x);
    const int deg = e1 - e0;
    {   // column ids of the row's first kEll neighbours (zeros past the row's end: their offsets are out of range anyway).
        // This second dependent fetch runs under the first MFMA groups; a padded per-batch neighbour table that would have
        // delivered the ids with the first round trip was built and measured: no difference (227.5 vs 228.3 k graphs/s on MIX).
        const __amdgpu_buffer_rsrc_t colr = slab_rsrc(col);
#pragma unroll
        for (int k = 0; k < kEll; ++k)
            nid[k] = __builtin_amdgcn_raw_buffer_load_b32(colr, k < deg ? (unsigned)(e0 + k) * 4u : kOob, 0, 0);
    }
    float ns[BWD ? kEll : 1];       // backward: 1 / deg of the neighbour the gradient row comes from
    if constexpr (BWD) {
        const __amdgpu_buffer_rsrc_t ir = slab_rsrc(invdeg);
#pragma unroll
        for (int k = 0; k < kEll; ++k)
            ns[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ir, k < deg ? (unsigned)nid[k] * 4u : kOob, 0, 0));
    }
    using GS = GatherSched<NT>;
    using GL = GatherLds<NT>;
    constexpr int kFillGaps = RL::on ? GL::kGaps : GS::kGaps;
    unsigned noff[kEll];             // GLOBAL byte offset of neighbour k's row slot (out of range: nothing to fetch -> zeros)
    unsigned loff[RL::on ? kEll / 2 : 1];       // LDS byte offsets of the slots, two u16 per register (zero row: not in the block)
    unsigned gneed = 0;              // wave-uniform: bit k = some lane of the wave fetches slot k from global memory
    if constexpr (RL::on) {
        const unsigned blk0 = blockIdx.x * 128u;
#pragma unroll
        for (int k = 0; k < kEll / 2; ++k) loff[k] = 0u;
#pragma unroll
        for (int k = 0; k < kEll; ++k) {
            const unsigned loc = (unsigned)nid[k] - blk0;
            const bool have = k < deg, inb = have && loc < 128u;
            loff[k >> 1] |= ((inb ? loc : 128u) * (unsigned)(RL::XS * 4) + 16u * g) << (16 * (k & 1));
            noff[k] = (have && !inb) ? (unsigned)nid[k] * (unsigned)(HP * 4) + 16u * g : kOob;
            gneed |= (__ballot(have && !inb) != 0ull ? 1u : 0u) << k;
        }
        gneed = __builtin_amdgcn_readfirstlane(gneed);
    } else {
#pragma unroll
        for (int k = 0; k < kEll; ++k) noff[k] = k < deg ? (unsigned)nid[k] * (unsigned)(HP * 4) + 16u * g : kOob;
    }
    int wmax = deg < kEll ? deg : kEll;          // wave-uniform number of neighbour slots anybody in the wave uses
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) wmax = max(wmax, __shfl_xor(wmax, o));
    wmax = __builtin_amdgcn_readfirstlane(wmax);
    f32x4 tb[RL::on ? 3 : GS::kWin][NT];         // LDS path: [0..1] global landing ring, [2] LDS landing buffer
    auto filler_lds = [&](auto qq) {
        constexpr int Q = decltype(qq)::value;
        const char* lbase = reinterpret_cast<const char*>(rowsl);
        static_for_<0, kEll>([&](auto kk) {       // adds first (a gap's adds precede its loads: the rings rely on it)
            constexpr int k = decltype(kk)::value;
            if constexpr (GL::add_gap(k) == Q) {
                if (k < wmax) {
#pragma unroll
                    for (int c = 0; c < NT; ++c) {
                        if constexpr (BWD) ag[c] += tb[2][c] * ns[k];
                        else ag[c] += tb[2][c];
                    }
                }
            }
            if constexpr (GL::gadd_gap(k) == Q) {
                if (gneed & (1u << k)) {
#pragma unroll
                    for (int c = 0; c < NT; ++c) {
                        if constexpr (BWD) ag[c] += tb[k % 2][c] * ns[k];
                        else ag[c] += tb[k % 2][c];
                    }
                }
            }
        });
        static_for_<0, kEll>([&](auto kk) {
            constexpr int k = decltype(kk)::value;
            if constexpr (GL::rd_gap(k) == Q) {
                if (k < wmax) {
                    const unsigned lo = (k & 1) ? (loff[k >> 1] >> 16) : (loff[k >> 1] & 0xffffu);
                    const f32x4* lr = reinterpret_cast<const f32x4*>(lbase + lo);
#pragma unroll
                    for (int c = 0; c < NT; ++c) tb[2][c] = lr[4 * c];
                }
                if (gneed & (1u << k)) {
#pragma unroll
                    for (int c = 0; c < NT; ++c) tb[k % 2][c] = buf_load(xr_, noff[k] + 64 * c);
                }
            }
        });
    };
    auto filler_glb = [&](auto qq) {
        constexpr int Q = decltype(qq)::value;            // gap index behind the (c, t) group c * NT + t of the self half
        static_for_<0, kEll>([&](auto kk) {
            constexpr int k = decltype(kk)::value;
            if constexpr (GS::add_gap(k) == Q) {
                if (k < wmax) {
#pragma unroll
                    for (int c = 0; c < NT; ++c) {
                        if constexpr (BWD) ag[c] += tb[k % GS::kWin][c] * ns[k];
                        else ag[c] += tb[k % GS::kWin][c];
                    }
                }
            }
        });
        static_for_<Q * GS::kP, (Q + 1) * GS::kP < kEll * NT ? (Q + 1) * GS::kP : kEll * NT>([&](auto ii) {
            constexpr int i = decltype(ii)::value, k = i / NT, c = i % NT;
            if (k < wmax) tb[k % GS::kWin][c] = buf_load(xr_, noff[k] + 64 * c);
        });
    };
    auto filler = [&](auto qq) {
        if constexpr (RL::on) filler_lds(qq);
        else filler_glb(qq);
    };

    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // one K-half in the round-robin group order (MfmaSeq); `base` = first fragment of the half in LDS, `rows` = its row operand
    auto contract_rr = [&](const f32x4* __restrict__ base, const f32x4 (&rows)[NT], auto&& fill) {
        using MS = MfmaSeq<NT>;
        f32x4 fr[4];
#pragma unroll
        for (int i = 0; i < 3; ++i) fr[i] = base[i * 64 + lane];
        static_for_<0, MS::kGroups>([&](auto gg) {
            constexpr int gi = decltype(gg)::value, n = MS::group_size(gi), u0 = 3 * gi;
            // the fourth member of the LAST group uses a register set of its own: requested a whole group ahead
            if constexpr (gi + 2 == MS::kGroups && MS::group_size(gi + 1) == 4) fr[3] = base[(u0 + 6) * 64 + lane];
            if constexpr (MS::kGroups == 1 && n == 4) fr[3] = base[3 * 64 + lane];
            static_for_<0, 4 * n>([&](auto pp) {
                constexpr int pos = decltype(pp)::value, j = pos / n, i = pos % n, u = u0 + i, c = u / NT, t = u % NT;
                constexpr int sl = 4 * u0 + pos;                    // slot index within the half
                acc[t] = mfma16x16x4(fr[i][j], rows[c][j], acc[t]);
                // behind a unit's last MFMA its fragment registers take the same member of the next group (an MFMA reads its
                // operands at issue); that member's first MFMA is n slots away, with a gap in between
                if constexpr (j == 3 && gi + 1 < MS::kGroups && i < 3) fr[i] = base[(u0 + 3 + i) * 64 + lane];
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (sl % 4 == 3) {
                    fill(std::integral_constant<int, sl / 4>{});
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
        });
    };
    if constexpr (NT >= 3) {
        // (tried in round 3, 1.038 -> 1.075 ms on MIX: waves 4-7 gathering FIRST, loads and adds only under their partners'
        // MFMA streams, then both K-halves as one MFMA stream -- a wave's non-MFMA work crawls under its partner's MFMAs)
        // (also tried, 1.037 -> 1.089 ms: EVERY wave gathering first -- no MFMA stream anywhere on the CU to crawl under -- and
        // both K-halves as pure MFMA streams afterwards: back to back the sixteen slots expose one LDS / L2 round trip each,
        // ~9 k ticks that the MFMA groups otherwise cover)
        contract_rr(wlds + NT * NT * 64, xs, filler);
        static_for_<MfmaSeq<NT>::kGaps, kFillGaps>([&](auto qq) { filler(qq); __builtin_amdgcn_sched_barrier(0); });
    } else {
        // one or two tiles: too few accumulators to stagger; the dependent chain is waited out before anything is issued
        // behind a unit (the fused kernels' mfma_drain, 2 x 40 cycles per link)
        static_for_<0, NT>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            static_for_<0, NT>([&](auto tt) {
                constexpr int t = decltype(tt)::value;
                const f32x4 b = wlds[((NT + c) * NT + t) * 64 + lane];
                acc[t] = mfma16x16x4(b[0], xs[c][0], acc[t]);
                acc[t] = mfma16x16x4(b[1], xs[c][1], acc[t]);
                acc[t] = mfma16x16x4(b[2], xs[c][2], acc[t]);
                acc[t] = mfma16x16x4(b[3], xs[c][3], acc[t]);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
                filler(std::integral_constant<int, c * NT + t>{});
                __builtin_amdgcn_sched_barrier(0);
            });
        });
        static_for_<NT * NT, kFillGaps>([&](auto qq) { filler(qq); __builtin_amdgcn_sched_barrier(0); });     // (narrow layers: the schedule outlasts the MFMA groups)
    }
    LSTAMP(K, 2);
    f32x4 ym[BWD ? NT : 1];
    if constexpr (BWD) {               // the mask rows land under the aggregate half
        const __amdgpu_buffer_rsrc_t yr_ = slab_rsrc(ymask);
        const unsigned off = (valid && ymask) ? (unsigned)row * (unsigned)(HP * 4) + 16u * g : kOob;
#pragma unroll
        for (int c = 0; c < NT; ++c) ym[c] = buf_load(yr_, off + 64 * c);
    }
    if (valid) {
        if (deg > kEll)                // the rest of a long row (the two terminals of a board; late in a game many rows), from the CSR
            long_row_tail<NT, BWD, false, 2, 4>(col, slab_rsrc(invdeg), slab_rsrc(x), e0 + kEll, e1, g, ag);
        if constexpr (!BWD) {
#pragma unroll
            for (int c = 0; c < NT; ++c) ag[c] *= sc;
            if (agg_out) {
                f32x4* ar = reinterpret_cast<f32x4*>(agg_out + (size_t)row * HP) + g;
#pragma unroll
                for (int c = 0; c < NT; ++c) ar[4 * c] = ag[c];
            }
        }
    }
    if constexpr (NT >= 3) {
        contract_rr(wlds, ag, [](auto) {});
    } else {
#pragma unroll
        for (int c = 0; c < NT; ++c) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const f32x4 b = wlds[(c * NT + t) * 64 + lane];
                acc[t] = mfma16x16x4(b[0], ag[c][0], acc[t]);
                acc[t] = mfma16x16x4(b[1], ag[c][1], acc[t]);
                acc[t] = mfma16x16x4(b[2], ag[c][2], acc[t]);
                acc[t] = mfma16x16x4(b[3], ag[c][3], acc[t]);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
            }
        }
    }
    // epilogue.  Operands are swapped (a = packed W^T fragment, b = the row fragment), so the MFMA computes the
    // TRANSPOSED tile: lane (r,g) holds out[row0+r][16t+4g .. 16t+4g+3] -- the same lane layout as the input rows.
    LSTAMP(K, 4);
    if (valid) {
        f32x4* yr = reinterpret_cast<f32x4*>(out + (size_t)row * HP) + g;
        if constexpr (!BWD) {
            const f32x4* br = reinterpret_cast<const f32x4*>(bias) + g;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                f32x4 v = acc[t] + br[4 * t];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = (v[q] > 0.f || !relu) ? v[q] : 0.f;
                yr[4 * t] = v;
            }
        } else {
            // (backward: agg_out, when given, is the TAP -- the same rows BEFORE the mask, final_conv_grads of the model)
            if (agg_out) {
                f32x4* tr = reinterpret_cast<f32x4*>(agg_out + (size_t)row * HP) + g;
#pragma unroll
                for (int t = 0; t < NT; ++t) tr[4 * t] = acc[t];
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                f32x4 v = acc[t];
                if (ymask) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = ym[t][q] > 0.f ? v[q] : 0.f;
                }
                yr[4 * t] = v;
            }
        }
    }
    LSTAMP(K, 5);
}

template <int NT>
__global__ __launch_bounds__(512) void sage_hidden_fwd_kernel(
    int n, const int* __restrict__ rowptr, const int* __restrict__ col,
    const float* __restrict__ invdeg, const float* __restrict__ x, const f32x4* __restrict__ wpack,
    const float* __restrict__ bias, float* __restrict__ y, float* __restrict__ agg_out, int relu) {
    extern __shared__ f32x4 wlds[];  // [2NT][NT][64]: W_l part, then W_r part
    sage_layer_body<NT, false>(n, rowptr, col, invdeg, x, wpack, bias, nullptr, y, agg_out, relu, wlds);
}

// G_l rows in, dY = [sum_T G / deg | G] [W_l ; W_r] masked by y_{l-1} (ymask, null: unmasked) out
template <int NT>
__global__ __launch_bounds__(512) void sage_hidden_bwd_kernel(
    int n, const int* __restrict__ rowptr_t, const int* __restrict__ col_t,
    const float* __restrict__ invdeg, const float* __restrict__ g_in, const f32x4* __restrict__ wpackb,
    const float* __restrict__ ymask, float* __restrict__ out, float* __restrict__ tap) {
    extern __shared__ f32x4 wlds[];  // [2][NT][NT][64]: W_l part, W_r part
    sage_layer_body<NT, true>(n, rowptr_t, col_t, invdeg, g_in, wpackb, nullptr, ymask, out, tap, 1, wlds);
}

// ---- ALL hidden layers of a stack in ONE launch (round 3) ---------------------------------------------------------------
// The per-layer launches above pay, for every layer, the launch itself plus a prologue in which nothing computes: weights,
// the block's own rows and the CSR row bounds arrive (one round trip), then the column ids (a second, dependent one) --
// 9 k of a layer's 45 k ticks on MIX (profiles/r03/layer_stamps_MIX_r03.txt).  When the whole batch fits ONE resident
// workgroup per CU (n <= 128 x CUs) the layers run as a loop inside one kernel instead:
//   * the CSR state of a row (neighbour offsets into LDS / global memory, 1/deg, the wave's slot count) is layer-invariant:
//     fetched ONCE;
//   * a wave's output rows are the next layer's self rows IN THE SAME LANE LAYOUT: they stay in registers, and go to the LDS
//     row copy (the neighbours inside the block) without touching memory;
//   * the next layer's weights are requested by LDS-DMA BEFORE the grid-wide barrier and land while the workgroup waits in it;
//   * only neighbour rows owned by OTHER workgroups come from global memory (the saved activations every layer writes
//     anyway), which is what the barrier between two layers is for.
// No grid-wide barrier: every 128-row block has a progress counter (its waves add 1 each per finished layer, after their
// stores are acknowledged); a wave that needs rows of OTHER blocks -- known from the row's neighbour list, layer-invariant --
// waits at the start of a layer until the blocks it reads from have finished the previous one.  Blocks made of whole graphs
// never wait; a graph cut by a block boundary couples just the blocks it touches, so the skew between workgroups does not
// add up over the layers the way it does with a kernel boundary (or a grid barrier: 1.10 ms per MIX step, against 1.02 ms
// with per-layer launches) after every layer.  Rows that cross workgroups go through AGENT-scope accesses (sc1: stores write
// through the XCD's L2, loads do not hit stale lines in it -- the eight L2s are not coherent with each other); fencing instead
// (buffer_wbl2 / buffer_inv per workgroup and layer) cost 1.37 ms per step.  Every workgroup is resident (host-side guard), so
// every wait ends; a poll budget (seconds) turns a would-be hang into HEXGNN_ETIMEOUT in the library's status word.
struct StackKArgs {
    int n, l_first, n_layers;          // forward: layers l_first, l_first + 1, ...; backward: l_first, l_first - 1, ...
    int relu_last, last_of_stack;      // forward: the stack's last layer index and whether it has a ReLU
    int tap_layer;                     // backward: layer whose output gradient (unmasked) also goes to tap_out (-1: none)
    const int* rowptr;                 // backward: the transposed CSR
    const int* col;
    const float* invdeg;
    const float* in0;                  // rows entering the first processed layer
    float* slabs;                      // forward: acts (layer l's output = slabs + slab * l); backward: G (output of layer l's
    size_t slab;                       //          launch = slabs + slab * (l - 1))
    const float* masks;                // backward: acts (mask of layer l's output gradient = masks + slab * (l - 1))
    float* dx;                         // backward: output of layer 0 (a hidden-width stack input)
    const char* w0;                    // packed weights of the first processed layer, wstride bytes per layer
    size_t wstride;
    const char* b0;                    // forward: bias of the first processed layer (same stride)
    char* agg0;                        // forward: saved aggregate of the first processed layer (null: not saved), astride per layer
    size_t astride;
    float* tap_out;
    unsigned* flags;                   // [blocks] progress counters, zero at launch
    const int* bstart;                 // null: block b = rows [128 b, 128 b + 128); else [nblocks + 1] row offsets (block b =
    int nblocks;                       //   rows [bstart[b], bstart[b + 1]), at most 128 each, a partition of [0, n))
    int* status;
    unsigned skew;                     // test aid (HEXGNN_STACK_SKEW): != 0 delays every block by a pseudo-random time per layer;
};                                     // 0xDE00bbbb: block bbbb never publishes its progress (its readers must time out)
constexpr unsigned kBarMaxPolls = 1u << 21;

constexpr int kAuxSc1 = 16;
__device__ __forceinline__ f32x4 buf_load_coh(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, kAuxSc1));
}
__device__ __forceinline__ void buf_store_coh(const f32x4 v, __amdgpu_buffer_rsrc_t r, unsigned off) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4b, v), r, off, 0, kAuxSc1);
}
// wave-wide wait: blocks lo..hi except `self` have all finished `target / 8` layers (lane t polls block lo + t, + 64, ...).
// The poll is a relaxed agent-scope load (global_load_dword sc1); the rows it guards are then read by THIS wave with sc1
// loads, after its poll has matched (MI355X guide, inter-workgroup visibility, first row of the sc1-loads table: one lane of
// each storing workgroup signals for all its stores -- see the publish points below).  Returns true when the poll budget
// ran out (never expected: every workgroup is resident): the caller then poisons what it stores, so that the call's output
// cannot pass for a result, and the status word says HEXGNN_ETIMEOUT.
__device__ __forceinline__ bool wait_blocks(const unsigned* flags, int lo, int hi, int self, unsigned target, int* status) {
    const int lane = threadIdx.x & 63;
    bool timed_out = false;
    for (int b0 = lo; b0 <= hi; b0 += 64) {
        const int j = b0 + lane;
        const bool mine = j <= hi && j != self;
        unsigned polls = 0;
        while (true) {
            const unsigned v = mine ? __hip_atomic_load(flags + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : target;
            if (__ballot((int)(v - target) < 0) == 0ull) break;          // (counters are compared modulo 2^32)
            __builtin_amdgcn_s_sleep(4);
            if (++polls > kBarMaxPolls) {
                if (status && lane == 0) *status = HEXGNN_ETIMEOUT;
                timed_out = true;
                break;
            }
        }
    }
    // the relaxed poll orders nothing for the compiler: keep every later load (the other blocks' rows) behind the loop
    asm volatile("" ::: "memory");
    return timed_out;
}

template <int NT, bool BWD>
__device__ __forceinline__ void sage_stack_body(const StackKArgs& a, f32x4* wlds) {
    static_assert(NT >= 3, "the round-robin MFMA order needs three tiles");
    constexpr int HP = 16 * NT;
    const int n = a.n;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_w = (unsigned)(size_t)(__attribute__((address_space(3))) char*)wlds;
    auto stage_weights = [&](const char* wp) {
        const f32x4* w4 = reinterpret_cast<const f32x4*>(wp);
        for (int p = wave; p < 2 * NT * NT; p += 8) dma_piece(w4 + p * 64, 16 * lane, lds_w + p * 1024);
    };
    // the block's rows.  With a block table (graph-aligned blocks: hexgnn_sage_stack_forward_blocks) the range comes from the
    // table and is checked HERE (the table is device data the host never saw): a range that is not a piece of a partition of
    // [0, n) in pieces of at most 128 rows makes the block empty and sets HEXGNN_EINVAL in the status word
    int brow0 = blockIdx.x * 128, bcnt = min(128, n - brow0);
    if (a.bstart) {
        const int nbk = (int)gridDim.x;
        brow0 = __builtin_amdgcn_readfirstlane(a.bstart[blockIdx.x]);
        const int bend = __builtin_amdgcn_readfirstlane(a.bstart[blockIdx.x + 1]);
        bcnt = bend - brow0;
        const bool ok = brow0 >= 0 && bcnt >= 0 && bcnt <= 128 && bend <= n && (blockIdx.x != 0 || brow0 == 0) &&
                        ((int)blockIdx.x != nbk - 1 || bend == n);
        if (!ok) {
            if (a.status && tid == 0) *a.status = HEXGNN_EINVAL;
            brow0 = 0; bcnt = 0;          // (stays in the protocol: a reader of the rows it should have owned must not time out)
        } else if (bcnt == 0) {
            // a VALID empty block (a table built on the device has as many entries as the grid: the unused ones sit at the end with
            // start == n) leaves before anything is in flight; nobody ever waits for it -- no row lies in its range
            return;
        }
    }
    stage_weights(a.w0);
    const int row0 = brow0 + wave * 16;
    const int r = lane & 15, g = lane >> 4;
    const int row = row0 + r;
    const bool valid = wave * 16 + r < bcnt;
    // a wave without rows (blocks shorter than 113 rows: packed batches, the last block) issues no MFMAs -- it would only take
    // the matrix pipe from the wave it shares its SIMD with -- but keeps its part in the hand-over (counters, staging, barriers)
    const bool wactive = wave * 16 < bcnt;
    f32x4 xs[NT], ag[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c) xs[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    int e0 = 0, e1 = 0;
    float sc = 0.f;                    // 1 / deg of the row: forward the scale of its aggregate, backward of its row as a SOURCE
    if (valid) {
        const f32x4* xr = reinterpret_cast<const f32x4*>(a.in0 + (size_t)row * HP) + g;
#pragma unroll
        for (int c = 0; c < NT; ++c) xs[c] = xr[4 * c];
        e0 = a.rowptr[row];
        e1 = a.rowptr[row + 1];
        sc = a.invdeg[row];
    }
    using RL = RowsLds<NT>;
    float* rowsl = reinterpret_cast<float*>(wlds + 2 * NT * NT * 64);
    // backward: a gathered row G_j enters the sum as G_j / deg_j -- a property of the SOURCE row, so the LDS copy holds the
    // rows already scaled and the in-block slots need no per-slot factor (sixteen registers less than the per-layer kernel)
    auto rows_to_lds = [&]() {
        if constexpr (RL::on) {
            f32x4* mine = reinterpret_cast<f32x4*>(rowsl + (wave * 16 + r) * RL::XS) + g;
#pragma unroll
            for (int c = 0; c < NT; ++c) {
                if constexpr (BWD) mine[4 * c] = xs[c] * sc;
                else mine[4 * c] = xs[c];
            }
        }
    };
    rows_to_lds();
    if constexpr (RL::on) {
        if (tid < RL::XS / 4) reinterpret_cast<f32x4*>(rowsl + 128 * RL::XS)[tid] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // With the LDS row copy (hidden 49..112) the hand-over between two layers keeps nothing on the critical path but two LDS
    // barriers (kV3).  Three workgroup counters behind the row copy:
    //   ctl[0]  waves that finished the self half (the W_r region is free once all eight have, while the slower waves are
    //           still in their aggregate half): the waves that are DONE with the layer -- they would only wait -- claim the
    //           next layer's W_r pieces one by one (ctl[1]) and stage them, so W_r is in place when the last wave arrives;
    //   ctl[2]  waves 0-3 stage the next layer's W_l part behind the barrier and report (a few MFMA groups into the next
    //           self half, where they wait for their pieces -- the other four waves keep the matrix pipe busy meanwhile);
    //           nobody enters an aggregate half before all four have.
    // The progress counter of the block (global) is raised at the same point, after the wave's stores are acknowledged,
    // and only later does a wave that reads other blocks' rows wait for those blocks (publishing first: two neighbouring
    // blocks wait for each other).
    // Rows of OTHER blocks are fetched in the TAIL of the self half (behind its last MFMA group; a ring of four landing
    // buffers: the three of the gather -- the LDS one is free by then -- and the registers of the self rows, dead there): a
    // neighbour block's counter needs a write-through acknowledge, an atomic and a poll round trip (8-10 k ticks after the
    // layer started, profiles/r03/stack_stamps_MIX_v3_remote_wait_at_gap3.txt); waiting for it a few groups into the self half
    // made the edge waves of every block the slow ones of every layer.  W_l and the bias are requested at the TOP of the next
    // layer (behind the second barrier), not between the barriers.
    constexpr bool kV3 = RL::on && !BWD;      // forward: + the remote rows in the tail of the self half, the publish in the hook
    // round 4: the hand-over itself (ctl[0] / ctl[2], W_r requested from inside the aggregate half, ONE barrier) is written for
    // both directions -- the backward would keep its remote rows where they are (waited for at the top of a layer, fetched in
    // line with the LDS slots: no registers for a tail ring) and publish at the END of a layer, behind its own stores'
    // acknowledge -- but pays only in the forward:
    constexpr bool kHO = RL::on && !BWD;      // (measured with the backward on it too: 298 us per MIX launch against 287 plain)
    constexpr int kGo = kHO ? 4 : 0;              // gap of the publish hook + 1
    constexpr int kGt = kV3 ? (GatherLds<NT>::G > GatherLds<NT>::add_gap(kEll - 1) + 1 ? GatherLds<NT>::G
                                                                                     : GatherLds<NT>::add_gap(kEll - 1) + 1) : 0;   // first tail gap
    constexpr unsigned kCtlWaves = BWD ? 4u : 5u; // waves that stage something behind barrier 1 (W_l; forward: + the bias)
    unsigned* ctl = reinterpret_cast<unsigned*>(rowsl + 129 * RL::XS);
    float* bias_lds = reinterpret_cast<float*>(ctl + 16);         // forward: the layer's bias, 1 KiB (one LDS-DMA piece)
    const unsigned lds_b = lds_w + (unsigned)(2 * NT * NT * 1024 + 129 * RL::XS * 4 + 64);
    if constexpr (kHO) {
        if (tid == 0) { ctl[0] = 0u; ctl[1] = 0u; ctl[2] = kCtlWaves; ctl[3] = 0u; }
        if constexpr (!BWD) {
            if (wave == 4) dma_piece(a.b0, 16 * lane, lds_b);     // (reads past the 4 * HP bias bytes, inside the pack buffer)
        }
    }
    auto lds_count = [&](int i) { return __hip_atomic_load(ctl + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto lds_bump = [&](int i) { if (lane == 0) __hip_atomic_fetch_add(ctl + i, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    wait_vmem();
    __syncthreads();

    // ---- layer-invariant CSR state of the row ----
    const int deg = e1 - e0;
    using GS = GatherSched<NT>;
    using GL = GatherLds<NT>;
    constexpr int kRing = 4;                      // landing buffers of the tail: tb[0..2] and the registers of the self rows (dead there)
    constexpr int kFillGaps = RL::on ? (kV3 ? kGt + kEll + kRing - 1 : GL::kGaps) : GS::kGaps;
    unsigned noff[kEll];
    unsigned loff[RL::on ? kEll / 2 : 1];
    unsigned gneed = 0;
    constexpr bool kNs = BWD && !RL::on;          // per-slot factors only where every neighbour comes from global memory
    float ns[kNs ? kEll : 1];
    const __amdgpu_buffer_rsrc_t ir_ = slab_rsrc(a.invdeg);
    const int blk = blockIdx.x;
    int dlo = blk, dhi = blk;                     // blocks this wave reads rows from
    {
        int nid[kEll];
        const __amdgpu_buffer_rsrc_t colr = slab_rsrc(a.col);
#pragma unroll
        for (int k = 0; k < kEll; ++k)
            nid[k] = __builtin_amdgcn_raw_buffer_load_b32(colr, k < deg ? (unsigned)(e0 + k) * 4u : kOob, 0, 0);
        if constexpr (kNs) {
#pragma unroll
            for (int k = 0; k < kEll; ++k)
                ns[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ir_, k < deg ? (unsigned)nid[k] * 4u : kOob, 0, 0));
        }
        // block that owns row j (only asked for rows OUTSIDE this block).  Table form: the two neighbouring blocks from
        // registers -- a graph cut by a block boundary continues in the next block --, anything else by bisection (every index
        // stays inside the table whatever it holds)
        int pb0 = 0, nb1 = 0, nb2 = 0;
        const int nbk = (int)gridDim.x;
        if (a.bstart) {
            pb0 = a.bstart[blk > 0 ? blk - 1 : 0];
            nb1 = a.bstart[blk + 1];
            nb2 = a.bstart[blk + 2 <= nbk ? blk + 2 : nbk];
        }
        auto block_of = [&](int j) -> int {
            if (!a.bstart) return j >> 7;
            if (j >= nb1 && j < nb2) return blk + 1 < nbk ? blk + 1 : blk;
            if (j >= pb0 && j < brow0) return blk > 0 ? blk - 1 : blk;
            int lo = 0, hi = nbk;
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (a.bstart[mid] <= j) lo = mid; else hi = mid;
            }
            return lo;
        };
#pragma unroll
        for (int k = 0; k < kEll; ++k) {
            if (k < deg && (unsigned)(nid[k] - brow0) >= (unsigned)bcnt) {
                const int j = block_of(nid[k]);
                dlo = min(dlo, j); dhi = max(dhi, j);
            }
        }
        if (valid) {
            for (int e = e0 + kEll; e < e1; ++e) {
                const int c = a.col[e];
                if ((unsigned)(c - brow0) < (unsigned)bcnt) continue;
                const int j = block_of(c);
                dlo = min(dlo, j); dhi = max(dhi, j);
            }
        }
        if constexpr (RL::on) {
            const unsigned blk0 = (unsigned)brow0;
#pragma unroll
            for (int k = 0; k < kEll / 2; ++k) loff[k] = 0u;
#pragma unroll
            for (int k = 0; k < kEll; ++k) {
                const unsigned loc = (unsigned)nid[k] - blk0;
                const bool have = k < deg, inb = have && loc < (unsigned)bcnt;
                loff[k >> 1] |= ((inb ? loc : 128u) * (unsigned)(RL::XS * 4) + 16u * g) << (16 * (k & 1));
                noff[k] = (have && !inb) ? (unsigned)nid[k] * (unsigned)(HP * 4) + 16u * g : kOob;
                gneed |= (__ballot(have && !inb) != 0ull ? 1u : 0u) << k;
            }
            gneed = __builtin_amdgcn_readfirstlane(gneed);
        } else {
#pragma unroll
            for (int k = 0; k < kEll; ++k) noff[k] = k < deg ? (unsigned)nid[k] * (unsigned)(HP * 4) + 16u * g : kOob;
        }
    }
    int wmax = deg < kEll ? deg : kEll;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        wmax = max(wmax, __shfl_xor(wmax, o));
        dlo = min(dlo, __shfl_xor(dlo, o));
        dhi = max(dhi, __shfl_xor(dhi, o));
    }
    wmax = __builtin_amdgcn_readfirstlane(wmax);
    dlo = __builtin_amdgcn_readfirstlane(dlo);
    dhi = __builtin_amdgcn_readfirstlane(dhi);
    // a row with more than kEll neighbours finishes its sum from GLOBAL memory, rows of its own block included: such a wave
    // also waits for its own block's counter (the other waves' stores acknowledged)
    const bool longrow = __ballot(deg > kEll) != 0ull;
    const int self_excl = longrow ? -1 : blk;
    const bool remote = dlo != blk || dhi != blk || longrow;        // wave-uniform
    // the counters are never reset between launches over the same pack buffer (a second backward over one forward): every
    // block ends a launch at the same value, 8 x (layers - 1) above where it started, so a block's own counter at kernel
    // start is everybody's starting value
    const unsigned fbase = __builtin_amdgcn_readfirstlane(
        __hip_atomic_load(a.flags + blk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    // (read by every wave before the workgroup's first publish: two workgroup barriers lie between this read and that add)
    asm volatile("" ::: "memory");

    // One signal per workgroup and layer, for ALL its stores: every wave drains its own stores (s_waitcnt vmcnt(0)), then adds to
    // an LDS counter, and the wave whose add is the eighth of the layer raises the block's global counter by 8 (the hand-over
    // without a barrier), or one lane does behind the workgroup barrier (the plain hand-over).  Until round 4 every wave added
    // 1 for itself right behind its own wait -- a form the guide's table lists only together with a workgroup barrier between the
    // consumer's poll and its loads and with whole 128-byte lines per store instruction (rows of 448 B: a store instruction
    // here writes 64 B per row).
    const bool muted = (a.skew >> 24) == 0xDEu && (int)(a.skew & 0xffffu) == blk;      // (test aid)
    auto publish_block = [&]() {
        asm volatile("" ::: "memory");
        if (lane == 0 && !muted) {
            const unsigned c = __hip_atomic_fetch_add(ctl + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if ((c & 7u) == 7u) __hip_atomic_fetch_add(a.flags + blk, 8u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    bool dead = false;                 // wave-uniform: a wait of this wave timed out -> everything it stores from now on is NaN
    const f32x4 kNan4 = f32x4{__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")};
    const float* xin = a.in0;
    for (int it = 0; it < a.n_layers; ++it) {
        const int l = BWD ? a.l_first - it : a.l_first + it;
        if (a.skew && (a.skew >> 24) != 0xDEu) {      // test aid: uneven progress of the blocks (stress test of the hand-over)
            unsigned h = (a.skew + 0x9e3779b9u * (unsigned)(blk + 1)) ^ (0x85ebca6bu * (unsigned)(it + 1));
            h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12;
            for (unsigned k = h & 63u; k > 0; --k) __builtin_amdgcn_s_sleep(64);
        }
        float* out = BWD ? (l >= 1 ? a.slabs + a.slab * (size_t)(l - 1) : a.dx) : a.slabs + a.slab * (size_t)l;
        const float* ymask = BWD ? (l >= 1 ? a.masks + a.slab * (size_t)(l - 1) : nullptr) : nullptr;
        float* side = BWD ? ((a.tap_out && l - 1 == a.tap_layer) ? a.tap_out : nullptr)
                          : (a.agg0 ? reinterpret_cast<float*>(a.agg0 + a.astride * (size_t)it) : nullptr);
        const float* bias = BWD ? nullptr : reinterpret_cast<const float*>(a.b0 + a.wstride * (size_t)it);
        const int relu = BWD ? 1 : ((l != a.last_of_stack) || a.relu_last);
        const float relu_lo = relu ? 0.f : -__builtin_inff();
        (void)relu_lo;
        const __amdgpu_buffer_rsrc_t xr_ = slab_rsrc(xin);
        constexpr int KS = BWD ? 1 : 0;
        (void)KS;
        PSTAMP(KS, 0);
        if constexpr (kHO) {
            // W_l (and the bias) of THIS layer: requested by waves 0-3 (4) behind the barrier, so that the other waves
            // are already in their self halves; needed from the aggregate half on (ctl[2])
            if (it > 0) {
                const f32x4* wc = reinterpret_cast<const f32x4*>(
                    a.w0 + (BWD ? -(ptrdiff_t)(a.wstride * (size_t)it) : (ptrdiff_t)(a.wstride * (size_t)it)));
                if (wave < 4) {
                    for (int pw = wave; pw < NT * NT; pw += 4) dma_piece(wc + pw * 64, 16 * lane, lds_w + pw * 1024);
                }
                if constexpr (!BWD) {
                    if (wave == 4) dma_piece(a.b0 + a.wstride * (size_t)it, 16 * lane, lds_b);
                }
            }
        }
        // (round 4, VALU diet: v_mfma_f32_16x16x4_f32 does not overlap VALU work on its SIMD, every VALU instruction is matrix
        // time lost.  With the LDS row copy the first neighbour slot lands in the sums directly: no zero fill, no `0 + x` add)
        if (!RL::on || wmax == 0) {
#pragma unroll
            for (int c = 0; c < NT; ++c) ag[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // landing buffers with the LDS row copy: [ring of the rows fetched from global memory | one for the LDS slots].  The
        // backward kernel sat at 256 VGPRs + 132 B of scratch per lane with a global ring of two: ONE buffer there (the remote
        // rows of a wave are few, and their load latency is exposed at the add either way), no spills
        constexpr int kGRing = kV3 ? 3 : (BWD ? 1 : 2);
        constexpr int kLb = RL::on ? (kV3 ? 2 : kGRing) : 0;      // the LDS slots' buffer (kV3: shared with the tail ring's third)
        f32x4 tb[RL::on ? (kV3 ? 3 : kGRing + 1) : GS::kWin][NT];
        float rs[kRing] = {0.f, 0.f, 0.f, 0.f};   // backward, LDS path: 1 / deg of the rows in the global landing ring
        (void)rs;
        if constexpr (!kV3) {
            if (it > 0 && remote && !dead) dead |= wait_blocks(a.flags, dlo, dhi, blk, fbase + 8u * (unsigned)it, a.status);
        }
        auto publish_hook = [&]() {
            if (it > 0) {
                if constexpr (kV3) {
                    wait_vmem();              // previous layer's rows written through; waves 0-3: their W_l pieces landed
                    if (wave < (int)kCtlWaves) lds_bump(2);
                    publish_block();
                } else {
                    if (wave < (int)kCtlWaves) { wait_vmem(); lds_bump(2); }      // (backward: published at the layer's end)
                }
            }
            PSTAMP(KS, 1);
        };
        auto filler_lds = [&](auto qq) {
            constexpr int Q = decltype(qq)::value;
            const char* lbase = reinterpret_cast<const char*>(rowsl);
            if constexpr (kHO && Q == kGo - 1) publish_hook();
            if constexpr (kV3 && Q == kGt) {
                if (it > 0 && remote && !dead) dead |= wait_blocks(a.flags, dlo, dhi, self_excl, fbase + 8u * (unsigned)it, a.status);
                PSTAMP(KS, 12);
            }
            static_for_<0, kEll>([&](auto kk) {
                constexpr int k = decltype(kk)::value;
                if constexpr (GL::add_gap(k) == Q && k > 0) {
                    if (k < wmax) {
                        asm volatile("" ::: "memory");        // (keeps hipcc from turning the block into "add, then select")
#pragma unroll
                        for (int c = 0; c < NT; ++c) ag[c] += tb[kLb][c];     // (backward: the LDS rows are pre-scaled)
                    }
                }
                if constexpr ((kV3 ? kGt + k + kRing - 1 : (BWD ? GL::rd_gap(k) + GL::stride : GL::gadd_gap(k))) == Q) {
                    constexpr int rb = kV3 ? k % kRing : k % kGRing;
                    if (gneed & (1u << k)) {
                        asm volatile("" ::: "memory");
#pragma unroll
                        for (int c = 0; c < NT; ++c) {
                            f32x4 v;
                            if constexpr (rb < 3) v = tb[rb][c];
                            else v = xs[c];
                            if constexpr (BWD) ag[c] += v * rs[rb];
                            else ag[c] += v;
                        }
                    }
                }
            });
            static_for_<0, kEll>([&](auto kk) {
                constexpr int k = decltype(kk)::value;
                if constexpr (GL::rd_gap(k) == Q) {
                    if (k < wmax) {
                        asm volatile("" ::: "memory");
                        const unsigned lo = (k & 1) ? (loff[k >> 1] >> 16) : (loff[k >> 1] & 0xffffu);
                        const f32x4* lr = reinterpret_cast<const f32x4*>(lbase + lo);
#pragma unroll
                        for (int c = 0; c < NT; ++c) {
                            if constexpr (k == 0) ag[c] = lr[4 * c];          // slot 0 (gap 0, ahead of every add): straight into the sums
                            else tb[kLb][c] = lr[4 * c];
                        }
                    }
                }
                if constexpr ((kV3 ? kGt + k : GL::rd_gap(k)) == Q) {
                    constexpr int rb = kV3 ? k % kRing : k % kGRing;
                    if (gneed & (1u << k)) {
#pragma unroll
                        for (int c = 0; c < NT; ++c) {
                            if constexpr (rb < 3) tb[rb][c] = buf_load_coh(xr_, noff[k] + 64 * c);
                            else xs[c] = buf_load_coh(xr_, noff[k] + 64 * c);
                        }
                        if constexpr (BWD) {      // 1 / deg of the remote source row: its id back from the byte offset
                            const unsigned j = (noff[k] - 16u * g) / (unsigned)(HP * 4);
                            rs[rb] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                ir_, noff[k] == kOob ? kOob : j * 4u, 0, 0));
                        }
                    }
                }
            });
        };
        auto filler_glb = [&](auto qq) {
            constexpr int Q = decltype(qq)::value;
            static_for_<0, kEll>([&](auto kk) {
                constexpr int k = decltype(kk)::value;
                if constexpr (GS::add_gap(k) == Q) {
                    if (k < wmax) {
#pragma unroll
                        for (int c = 0; c < NT; ++c) {
                            if constexpr (BWD) ag[c] += tb[k % GS::kWin][c] * ns[k];
                            else ag[c] += tb[k % GS::kWin][c];
                        }
                    }
                }
            });
            static_for_<Q * GS::kP, (Q + 1) * GS::kP < kEll * NT ? (Q + 1) * GS::kP : kEll * NT>([&](auto ii) {
                constexpr int i = decltype(ii)::value, k = i / NT, c = i % NT;
                if (k < wmax) tb[k % GS::kWin][c] = buf_load_coh(xr_, noff[k] + 64 * c);
            });
        };
        auto filler = [&](auto qq) {
            if constexpr (RL::on) filler_lds(qq);
            else filler_glb(qq);
        };
        f32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto contract_rr = [&](const f32x4* __restrict__ base, const f32x4 (&rows)[NT], auto&& fill) {
            using MS = MfmaSeq<NT>;
            f32x4 fr[4];
#pragma unroll
            for (int i = 0; i < 3; ++i) fr[i] = base[i * 64 + lane];
            static_for_<0, MS::kGroups>([&](auto gg) {
                constexpr int gi = decltype(gg)::value, nn = MS::group_size(gi), u0 = 3 * gi;
                if constexpr (gi + 2 == MS::kGroups && MS::group_size(gi + 1) == 4) fr[3] = base[(u0 + 6) * 64 + lane];
                if constexpr (MS::kGroups == 1 && nn == 4) fr[3] = base[3 * 64 + lane];
                static_for_<0, 4 * nn>([&](auto pp) {
                    constexpr int pos = decltype(pp)::value, j = pos / nn, i = pos % nn, u = u0 + i, c = u / NT, t = u % NT;
                    constexpr int sl = 4 * u0 + pos;
                    acc[t] = mfma16x16x4(fr[i][j], rows[c][j], acc[t]);
                    if constexpr (j == 3 && gi + 1 < MS::kGroups && i < 3) fr[i] = base[(u0 + 3 + i) * 64 + lane];
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (sl % 4 == 3) {
                        fill(std::integral_constant<int, sl / 4>{});
                        __builtin_amdgcn_sched_barrier(0);
                    }
                });
            });
        };
        if (wactive) {
            contract_rr(wlds + NT * NT * 64, xs, filler);
            static_for_<MfmaSeq<NT>::kGaps, kFillGaps>([&](auto qq) { filler(qq); __builtin_amdgcn_sched_barrier(0); });
        } else {
            if constexpr (kHO) publish_hook();
        }
        if constexpr (kHO) lds_bump(0);           // this wave's last W_r fragment has been read
        PSTAMP(KS, 2);
        f32x4 ym[BWD ? NT : 1];
        if constexpr (BWD) {
            const __amdgpu_buffer_rsrc_t yr_ = slab_rsrc(ymask);
            const unsigned off = (valid && ymask) ? (unsigned)row * (unsigned)(HP * 4) + 16u * g : kOob;
#pragma unroll
            for (int c = 0; c < NT; ++c) ym[c] = buf_load(yr_, off + 64 * c);
        }
        if (valid) {
            if (deg > kEll) long_row_tail<NT, BWD, true, BWD ? 1 : 2, HEXGNN_LR_IDS>(a.col, ir_, xr_, e0 + kEll, e1, g, ag);
            if constexpr (!BWD) {
#pragma unroll
                for (int c = 0; c < NT; ++c) ag[c] *= sc;
                if (side) {
                    f32x4* ar = reinterpret_cast<f32x4*>(side + (size_t)row * HP) + g;
#pragma unroll
                    for (int c = 0; c < NT; ++c) ar[4 * c] = ag[c];
                }
            }
        }
        PSTAMP(KS, 3);
        if constexpr (kHO) {
            while (lds_count(2) < kCtlWaves * (unsigned)(it + 1)) __builtin_amdgcn_s_sleep(1);   // (W_l / bias of this layer in place)
        }
        PSTAMP(KS, 4);
        // W_r of the NEXT layer: its LDS region is free once every wave is through its self half (ctl[0]), which is long before
        // this wave is through its aggregate half -- each wave requests its share of the pieces from inside the aggregate half
        // (a counter read at a few gaps; round 4: the pieces used to be claimed by the waves that were done with the layer and
        // landed 2-3 k ticks after the last epilogue, profiles/r04/stack_stamps_blocks.txt)
        bool wr_issued = !kHO || it + 1 == a.n_layers;
        auto issue_wr = [&]() {
            const f32x4* wn = reinterpret_cast<const f32x4*>(
                a.w0 + (BWD ? -(ptrdiff_t)(a.wstride * (size_t)(it + 1)) : (ptrdiff_t)(a.wstride * (size_t)(it + 1))));
            for (int pc = wave; pc < NT * NT; pc += 8) {
                const unsigned pw = (unsigned)(NT * NT + pc);
                dma_piece(wn + pw * 64, 16 * lane, lds_w + pw * 1024);
            }
            wr_issued = true;
        };
        auto filler_agg = [&](auto qq) {
            constexpr int Q = decltype(qq)::value;
            if constexpr (kHO && (Q == 1 || Q == MfmaSeq<NT>::kGaps / 3 || Q == 2 * MfmaSeq<NT>::kGaps / 3)) {
                if (!wr_issued && lds_count(0) >= 8u * (unsigned)(it + 1)) issue_wr();
            }
        };
        if (wactive) contract_rr(wlds, ag, filler_agg);
        if constexpr (kHO) {
            if (!wr_issued) {         // (a wave without rows, or one that was ahead of the others at every check)
                while (lds_count(0) < 8u * (unsigned)(it + 1)) __builtin_amdgcn_s_sleep(2);      // every self half is over
                issue_wr();
            }
            // the pieces have landed (requested a few thousand ticks ago) BEFORE the epilogue's stores are issued: nothing
            // waits for a write-through acknowledge on the way to the barrier
            wait_vmem();
        }
        PSTAMP(KS, 5);
        // epilogue: the stored rows ARE the next layer's self rows, in the same lane layout -> they stay in xs
        if (valid) {
            const __amdgpu_buffer_rsrc_t or_ = slab_rsrc(out);
            const unsigned oo = (unsigned)row * (unsigned)(HP * 4) + 16u * g;
            if constexpr (!BWD) {
                const f32x4* br = reinterpret_cast<const f32x4*>(kV3 ? bias_lds : bias) + g;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    f32x4 v = acc[t] + br[4 * t];
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], relu_lo);       // (one v_max; was compare + select)
                    if (dead) v = kNan4;
                    buf_store_coh(v, or_, oo + 64 * t);
                    xs[t] = v;
                }
            } else {
                if (side) {
                    f32x4* tr = reinterpret_cast<f32x4*>(side + (size_t)row * HP) + g;
#pragma unroll
                    for (int t = 0; t < NT; ++t) tr[4 * t] = acc[t];
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    f32x4 v = acc[t];
                    if (ymask) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] = ym[t][q] > 0.f ? v[q] : 0.f;
                    }
                    if (dead) v = kNan4;
                    buf_store_coh(v, or_, oo + 64 * t);
                    xs[t] = v;
                }
            }
        }
        PSTAMP(KS, 6);
        if (it + 1 == a.n_layers) break;
        // ---- between two layers ----
        if constexpr (kHO) {
            // every self half is over (this wave's W_r pieces could be requested): nobody reads the LDS row copy any more, this
            // wave's output rows go there now; ONE barrier: rows written, W_r landed, every aggregate half over (W_l may be
            // replaced: requested at the top of the next layer)
            rows_to_lds();
            if constexpr (BWD) {
                // the block's signal, behind this wave's acknowledged stores (the eighth arrival raises the global counter); and
                // every wave is past this point before anybody leaves the barrier: a long row may read its own block's rows
                wait_vmem();
                publish_block();
            }
            PSTAMP(KS, 8);
            lds_barrier();
            PSTAMP(KS, 9);
            xin = out;
            continue;
        }
        lds_barrier();        // every wave is past its MFMAs and its reads of the LDS rows (no wait for the stores here)
        rows_to_lds();
        stage_weights(a.w0 + (BWD ? -(ptrdiff_t)(a.wstride * (size_t)(it + 1)) : (ptrdiff_t)(a.wstride * (size_t)(it + 1))));
        wait_vmem();          // this wave's rows are written through, its weight pieces have landed
        __syncthreads();
        // ONE lane signals for the whole workgroup, behind the barrier that follows every wave's drained stores
        if (tid == 0 && !muted) __hip_atomic_fetch_add(a.flags + blk, 8u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        xin = out;
    }
}

template <int NT>
__global__ __launch_bounds__(512) void sage_stack_fwd_kernel(StackKArgs a) {
    extern __shared__ f32x4 wlds[];
    if constexpr (NT >= 3) sage_stack_body<NT, false>(a, wlds);
}
template <int NT>
__global__ __launch_bounds__(512) void sage_stack_bwd_kernel(StackKArgs a) {
    extern __shared__ f32x4 wlds[];
    if constexpr (NT >= 3) sage_stack_body<NT, true>(a, wlds);
}

// ---- out = dxs + sum_{j in T(i)} dagg_j, optionally masked by y>0 (stack-input gradient / G of a raw first layer) ----
__global__ void sage_combine_kernel(int n, int hp, const int* __restrict__ rowptr_t, const int* __restrict__ col_t,
                                    const float* __restrict__ dxs, const float* __restrict__ dagg,
                                    const float* __restrict__ ymask, float* __restrict__ out) {
    const int q4 = hp / 4;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n * q4) return;
    const int row = (int)(i / q4), p = (int)(i % q4);
    f32x4 v = reinterpret_cast<const f32x4*>(dxs + (size_t)row * hp)[p];
    if (dagg) {
        for (int e = rowptr_t[row]; e < rowptr_t[row + 1]; ++e)
            v += reinterpret_cast<const f32x4*>(dagg + (size_t)col_t[e] * hp)[p];
    }
    if (ymask) {
        const f32x4 yv = reinterpret_cast<const f32x4*>(ymask + (size_t)row * hp)[p];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = yv[j] > 0.f ? v[j] : 0.f;
    }
    reinterpret_cast<f32x4*>(out + (size_t)row * hp)[p] = v;
}

// ---- batched weight gradient (exact fp32): sage_dw_kernel.h -----------------------------------------------

// ---- the same batched weight gradient in split precision ("f16x3", math 1) ---------------------------------
// The contraction runs over ROWS, so both MFMA operands need 8 consecutive rows of one column per lane.  Every thread
// stages TWO rows (i, i+16) of one float4 column group (any fixed pairing works: the contraction index is permuted the
// same way for both operands, and this one keeps every wave-wide global load contiguous): the chunk (R = 32 rows = one K step) is scaled by the
// layer's power of two (max |G_l|, max |[agg|x]| -> 2^14..2^15, maxima produced by the fused forward / backward
// kernels), split into fp16 hi / lo and stored as ROW-PAIR words (row i in the low half, row i+16 in the high half)
// in two planes T[plane][rowpair][col].  A fragment is then four conflict-free ds_read_b32 per plane (row-pair stride
// == 4 mod 8 dwords) with no unpacking at all.  Wave w owns the input-feature tiles {2w, 2w+1} of [agg | x] and all
// NT output tiles: 9 fragments feed 42 v_mfma_f32_16x16x32_f16 per chunk.  db is summed exactly in fp32 on the staging
// path.  Same slab layout as sage_dw_kernel (deterministic slice reduce afterwards).
struct Dw16Args {
    const float* xin[kMaxLayers];
    const float* agg[kMaxLayers];
    const float* g[kMaxLayers];
    const unsigned* xmax;     // [absolute layer] bit pattern of max |[agg | x]|
    const unsigned* gmax;     // [absolute layer] bit pattern of max |G|
    int first_hidden, n, rows_per_slice, S;
};
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// rows (a, b) of one column, scaled: hi word = (f16(a s), f16(b s)), lo word = the fp16 remainders
__device__ __forceinline__ void split_rowpair(float va, float vb, float sa, float sb, unsigned& hi, unsigned& lo) {
    const float a = va * sa, b = vb * sb;
    const h16x2 h = {(_Float16)a, (_Float16)b};
    const h16x2 l = {(_Float16)(a - (float)h[0]), (_Float16)(b - (float)h[1])};
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}
__device__ __forceinline__ h16x8 dw16_frag(const unsigned* __restrict__ base /* &T[4kq][col] */, int stride) {
    return __builtin_bit_cast(h16x8, (u32x4){base[0], base[stride], base[2 * stride], base[3 * stride]});
}

template <int NT>
__global__ __launch_bounds__(64 * NT) void sage_dw16_kernel(Dw16Args a, float* __restrict__ part) {
    constexpr int HP = 16 * NT, R = 32, Q = 4 * NT, NTHR = 64 * NT;
    constexpr int XS2 = 2 * HP + 4;     // dwords per row pair; 4*XS2 == 16 (mod 32): conflict-free fragment reads
    constexpr int GS2 = HP + 4;
    constexpr int RP = R / 2;           // row pairs per chunk
    static_assert(RP * Q == NTHR, "one (row pair, column group) per thread");
    __shared__ __attribute__((aligned(16))) unsigned Xh[RP * XS2], Xl[RP * XS2];
    __shared__ __attribute__((aligned(16))) unsigned Gh[RP * GS2], Gl[RP * GS2];
    const int li = blockIdx.y, s = blockIdx.x;
    const int labs = a.first_hidden + li;
    const float* __restrict__ xin = a.xin[li];
    const float* __restrict__ agg = a.agg[li];
    const float* __restrict__ gg = a.g[li];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int r_beg = s * a.rows_per_slice;
    const int r_end = min(a.n, r_beg + a.rows_per_slice);
    float sx, ix, sg, ig;
    pow2_scale(__builtin_bit_cast(float, a.xmax[labs]), sx, ix);
    pow2_scale(__builtin_bit_cast(float, a.gmax[labs]), sg, ig);

    f32x4 acc[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t) { acc[t][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[t][1] = acc[t][0]; }
    f32x4 gsum = f32x4{0.f, 0.f, 0.f, 0.f};
    if (r_beg >= r_end) return;         // block-uniform (cannot happen with the host's slicing)

    const int q = tid % Q, rp = tid / Q;            // this thread's column group and row pair
    // two register sets of prefetched rows: the loads of chunk i+2 are issued while chunk i is multiplied (86 KB in flight
    // per CU; one workgroup per CU, <= 256 VGPRs)
    struct Pre { f32x4 ra[2], rx[2], rg[2]; float f0, f1; };   // f = 1 when the staged row exists (rows past the slice: zeros)
    Pre pA, pB;
    auto issue = [&](Pre& p, int rc) {
        const int row0 = rc + rp, row1 = row0 + RP;   // rows (i, i+16): both loads of a wave are contiguous 1 KB pieces
        p.f0 = row0 < r_end ? 1.f : 0.f; p.f1 = row1 < r_end ? 1.f : 0.f;
        const size_t o0 = (size_t)min(row0, r_end - 1) * HP, o1 = (size_t)min(row1, r_end - 1) * HP;
        p.ra[0] = reinterpret_cast<const f32x4*>(agg + o0)[q]; p.ra[1] = reinterpret_cast<const f32x4*>(agg + o1)[q];
        p.rx[0] = reinterpret_cast<const f32x4*>(xin + o0)[q]; p.rx[1] = reinterpret_cast<const f32x4*>(xin + o1)[q];
        p.rg[0] = reinterpret_cast<const f32x4*>(gg + o0)[q];  p.rg[1] = reinterpret_cast<const f32x4*>(gg + o1)[q];
    };
    auto stage = [&](const Pre& p) {
        // conditional adds, NOT `gsum += rg[0] * f0 + rg[1] * f1`: hipcc 7.2 turned that form into a
        // v_mul/v_pk_fma_f32 sequence whose third component came out ~6 % low on gfx950 (caught by the parity tests)
        if (p.f0 != 0.f) gsum += p.rg[0];
        if (p.f1 != 0.f) gsum += p.rg[1];
        const float sx0 = sx * p.f0, sx1 = sx * p.f1, sg0 = sg * p.f0, sg1 = sg * p.f1;
        u32x4 ah, al, xh, xl, gh, gl;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            unsigned h, l;
            split_rowpair(p.ra[0][c], p.ra[1][c], sx0, sx1, h, l); ah[c] = h; al[c] = l;
            split_rowpair(p.rx[0][c], p.rx[1][c], sx0, sx1, h, l); xh[c] = h; xl[c] = l;
            split_rowpair(p.rg[0][c], p.rg[1][c], sg0, sg1, h, l); gh[c] = h; gl[c] = l;
        }
        *reinterpret_cast<u32x4*>(&Xh[rp * XS2 + 4 * q]) = ah;
        *reinterpret_cast<u32x4*>(&Xl[rp * XS2 + 4 * q]) = al;
        *reinterpret_cast<u32x4*>(&Xh[rp * XS2 + HP + 4 * q]) = xh;
        *reinterpret_cast<u32x4*>(&Xl[rp * XS2 + HP + 4 * q]) = xl;
        *reinterpret_cast<u32x4*>(&Gh[rp * GS2 + 4 * q]) = gh;
        *reinterpret_cast<u32x4*>(&Gl[rp * GS2 + 4 * q]) = gl;
    };
    auto compute = [&]() {
        h16x8 bh[2], bl[2];
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
            const int o = (4 * kq) * XS2 + 16 * (2 * w + tb) + m;
            bh[tb] = dw16_frag(&Xh[o], XS2);
            bl[tb] = dw16_frag(&Xl[o], XS2);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int o = (4 * kq) * GS2 + 16 * t + m;
            const h16x8 ah = dw16_frag(&Gh[o], GS2);
            const h16x8 al = dw16_frag(&Gl[o], GS2);
            acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[0], acc[t][0], 0, 0, 0);
            acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[1], acc[t][1], 0, 0, 0);
            acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[0], acc[t][0], 0, 0, 0);
            acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[1], acc[t][1], 0, 0, 0);
            acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[0], acc[t][0], 0, 0, 0);
            acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[1], acc[t][1], 0, 0, 0);
        }
    };
    issue(pA, r_beg);
    issue(pB, r_beg + R);          // rows past the slice are clamped + flagged, so an over-issue is harmless
    for (int rc = r_beg; rc < r_end; rc += 2 * R) {
        stage(pA);
        __syncthreads();
        issue(pA, rc + 2 * R);
        compute();
        __syncthreads();
        if (rc + R < r_end) {       // block-uniform
            stage(pB);
            __syncthreads();
            issue(pB, rc + 3 * R);
            compute();
            __syncthreads();
        }
    }
    // slab [HP][2HP] then bias [HP]
    float* slab = part + ((size_t)li * a.S + s) * ((size_t)HP * (2 * HP + 1));
    const float inv = ix * ig;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                slab[(size_t)(16 * t + 4 * kq + r) * (2 * HP) + 16 * (2 * w + tb) + m] = acc[t][tb][r] * inv;
    float* red = reinterpret_cast<float*>(Xh);      // [16][HP] partial column sums of G (exact fp32)
    *reinterpret_cast<f32x4*>(&red[rp * HP + 4 * q]) = gsum;
    __syncthreads();
    if (tid < HP) {
        float b = 0.f;
#pragma unroll
        for (int k = 0; k < RP; ++k) b += red[k * HP + tid];
        slab[(size_t)HP * 2 * HP + tid] = b;
    }
}

struct DwReduceArgs {
    float* dwl[kMaxLayers];
    float* dbl[kMaxLayers];
    float* dwr[kMaxLayers];
    int S, hp, hidden;
};

// out element space per layer: [hidden][2*hidden + 1]; fixed summation order over slices (deterministic)
__device__ __forceinline__ void sage_dw_reduce_body(const DwReduceArgs& a, const float* __restrict__ part, int bx, int li) {
    const int H = a.hidden, hp = a.hp;
    const int idx = bx * 256 + (int)threadIdx.x;
    const int per = 2 * H + 1;
    if (idx >= H * per) return;
    const int o = idx / per, c = idx % per;
    const size_t slab_sz = (size_t)hp * (2 * hp + 1);
    size_t off;
    if (c < H) off = (size_t)o * 2 * hp + c;
    else if (c < 2 * H) off = (size_t)o * 2 * hp + hp + (c - H);
    else off = (size_t)hp * 2 * hp + o;
    const float* p = part + (size_t)li * a.S * slab_sz + off;
    float sum = 0.f;
    int s = 0;
    for (; s + 8 <= a.S; s += 8) {         // eight slabs' values requested together, added in slice order
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = p[(size_t)(s + j) * slab_sz];
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += v[j];
    }
    for (; s < a.S; ++s) sum += p[(size_t)s * slab_sz];
    if (c < H) a.dwl[li][o * H + c] = sum;
    else if (c < 2 * H) a.dwr[li][o * H + (c - H)] = sum;
    else a.dbl[li][o] = sum;
}
__global__ __launch_bounds__(256) void sage_dw_reduce_kernel(DwReduceArgs a, const float* __restrict__ part) {
    sage_dw_reduce_body(a, part, blockIdx.x, blockIdx.y);
}

// ---- raw first layer weight gradient: partial [S][hp][17] = sum_rows G[row][o] * (agg0[row][0..7] | x0[row][0..7] | 1) ----
__device__ __forceinline__ void sage_first_dw_body(
    int n, int c_in, int hp, int rows_per_slice /* <= 128 */, const float* __restrict__ g, const float* __restrict__ agg0,
    const float* __restrict__ x, int x_stride, float* __restrict__ part, int bx) {
    __shared__ float s_in[128][16];   // per row: agg0[0..7] | x0[0..7]
    __shared__ float red[128 * 17];
    const int tid = threadIdx.x, o = tid & 127, ph = tid >> 7;
    const int r_beg = bx * rows_per_slice, r_end = min(n, r_beg + rows_per_slice);
    const int rows = r_end - r_beg;
    for (int i = tid; i < 128 * 16; i += 256) {
        const int rr = i >> 4, q = i & 15;
        float v = 0.f;
        if (rr < rows) {
            if (q < kSmallCin) v = agg0[(size_t)(r_beg + rr) * kSmallCin + q];
            else if (q - kSmallCin < c_in) v = x[(size_t)(r_beg + rr) * x_stride + (q - kSmallCin)];
        }
        s_in[rr][q] = v;
    }
    __syncthreads();
    float acc[17];
#pragma unroll
    for (int q = 0; q < 17; ++q) acc[q] = 0.f;
    if (o < hp) {
        int rr = ph;
        for (; rr + 14 < rows; rr += 16) {          // eight rows' loads in flight, same summation order
            float gv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) gv[u] = g[(size_t)(r_beg + rr + 2 * u) * hp + o];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[q] += gv[u] * s_in[rr + 2 * u][q];
                acc[16] += gv[u];
            }
        }
        for (; rr < rows; rr += 2) {
            const float gv = g[(size_t)(r_beg + rr) * hp + o];
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] += gv * s_in[rr][q];
            acc[16] += gv;
        }
    }
    if (ph == 1) {
#pragma unroll
        for (int q = 0; q < 17; ++q) red[o * 17 + q] = acc[q];
    }
    __syncthreads();
    if (ph == 0 && o < hp) {
        float* out = part + ((size_t)bx * hp + o) * 17;
#pragma unroll
        for (int q = 0; q < 17; ++q) out[q] = acc[q] + red[o * 17 + q];
    }
}
__global__ __launch_bounds__(256) void sage_first_dw_kernel(
    int n, int c_in, int hp, int rows_per_slice, const float* __restrict__ g, const float* __restrict__ agg0,
    const float* __restrict__ x, int x_stride, float* __restrict__ part) {
    sage_first_dw_body(n, c_in, hp, rows_per_slice, g, agg0, x, x_stride, part, blockIdx.x);
}
// the slab reduce of the hidden layers and the raw first layer's partial sums are independent: ONE launch, workgroups
// [0, nrb * nh) reduce, the rest take one row slice of the first layer each
__global__ __launch_bounds__(256) void sage_dw_reduce_first_kernel(DwReduceArgs a, const float* __restrict__ part, int nrb, int nh,
                                                                  int n, int c_in, int rows_per_slice,
                                                                  const float* __restrict__ g0, const float* __restrict__ agg0,
                                                                  const float* __restrict__ x, int x_stride,
                                                                  float* __restrict__ part0) {
    const int bx = blockIdx.x;
    if (bx < nrb * nh) sage_dw_reduce_body(a, part, bx % nrb, bx / nrb);
    else sage_first_dw_body(n, c_in, a.hp, rows_per_slice, g0, agg0, x, x_stride, part0, bx - nrb * nh);
}

__global__ __launch_bounds__(64) void sage_first_dw_reduce_kernel(
    int S, int hp, int hidden, int c_in, const float* __restrict__ part, float* __restrict__ dwl,
    float* __restrict__ dbl, float* __restrict__ dwr) {
    // one wave per output element; lanes stride over the S partial slabs, fixed-shape tree => deterministic
    const int idx = blockIdx.x, lane = threadIdx.x;
    const int per = 2 * c_in + 1;
    const int o = idx / per, c = idx % per;
    const int q = c < c_in ? c : (c < 2 * c_in ? kSmallCin + (c - c_in) : 16);
    float sum = 0.f;
    for (int s = lane; s < S; s += 64) sum += part[((size_t)s * hp + o) * 17 + q];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
    if (lane == 0) {
        if (c < c_in) dwl[o * c_in + c] = sum;
        else if (c < 2 * c_in) dwr[o * c_in + (c - c_in)] = sum;
        else dbl[o] = sum;
    }
}

// ---- host-side dispatch ---------------------------------------------------------------------------------
template <int NT>
static void launch_fwd(int n, const int* rowptr, const int* col, const float* invdeg, const float* x,
                       const void* wp, const float* bias, float* y, float* agg, int relu, hipStream_t st) {
    static bool once = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sage_hidden_fwd_kernel<NT>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 2 * NT * NT * 1024 + RowsLds<NT>::bytes);
        return true;
    }();
    (void)once;
    KernelTimer kt(HEXGNN_K_SAGE_FWD, st);
    sage_hidden_fwd_kernel<NT><<<(n + 127) / 128, 512, 2 * NT * NT * 1024 + RowsLds<NT>::bytes, st>>>(
        n, rowptr, col, invdeg, x, (const f32x4*)wp, bias, y, agg, relu);
}

template <int NT>
static void launch_bwd(int n, const int* rowptr_t, const int* col_t, const float* invdeg,
                       const float* g_in, const void* wpb, const float* ymask, float* out, hipStream_t st,
                       float* tap = nullptr) {
    static bool once = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sage_hidden_bwd_kernel<NT>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 2 * NT * NT * 1024 + RowsLds<NT>::bytes);
        return true;
    }();
    (void)once;
    KernelTimer kt(HEXGNN_K_SAGE_BWD, st);
    sage_hidden_bwd_kernel<NT><<<(n + 127) / 128, 512, 2 * NT * NT * 1024 + RowsLds<NT>::bytes, st>>>(
        n, rowptr_t, col_t, invdeg, g_in, (const f32x4*)wpb, ymask, out, tap);
}

// One-launch stack kernels: usable when every workgroup can be resident at once (one per CU: 128 rows x CUs) and the status
// word (pinned host memory the kernels can write: a poll budget exceeded) exists.  HEXGNN_NO_PERSIST=1 keeps the per-layer
// launches (A/B measurements, debugging).
static int g_cu_count = 0;
static int* g_stack_status = nullptr;
static bool persist_ready(hipStream_t st) {
    static std::mutex mu;                      // (two host threads may issue their first stack call at the same time)
    static int state = 0;                      // 0 = not tried yet, 1 = ready, -1 = unavailable
    std::lock_guard<std::mutex> lock(mu);
    if (state != 0) return state > 0;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return false;                          // (no allocation while a graph is being captured: try again at the next call)
    }
    state = -1;
    const char* off = getenv("HEXGNN_NO_PERSIST");
    if (off && off[0] && off[0] != '0') return false;
    int dev = 0, cus = 0;
    void* p = nullptr;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        cus <= 0 || hipHostMalloc(&p, 64, hipHostMallocMapped) != hipSuccess || !p) {
        (void)hipGetLastError();
        return false;
    }
    g_stack_status = static_cast<int*>(p);
    *g_stack_status = 0;
    g_cu_count = cus;
    state = 1;
    return true;
}
// a poll budget exceeded in an EARLIER launch is reported by the next stack call (like hipGetLastError) or by
// hexgnn_stack_status() at any synchronisation point of the caller; the launch that timed out has poisoned its own output
// with NaN (sage_stack_body: `dead`), so its results cannot pass for valid in the meantime
static int take_stack_status() {
    if (!g_stack_status) return HEXGNN_OK;
    const int c = *reinterpret_cast<volatile int*>(g_stack_status);
    if (c != 0) *g_stack_status = 0;
    return c;
}
// ---- residency guard of the one-launch kernels ------------------------------------------------------------------------------
// Every workgroup must be resident at once (a wave polls other blocks' counters).  The launch is refused (-> per-layer launches)
// unless: the grid fits (occupancy x CUs - the CUs reserved for kernels that run beside it, e.g. RCCL channels while the
// gradient all-reduce overlaps the backward: hexgnn_stack_reserve_cus); no CU mask is in force; and no one-launch kernel of
// THIS process is still in flight on another stream (an event recorded behind every such launch; same-stream launches are
// ordered).  What it cannot see -- another process on the GPU, a kernel of another library that fills the CUs -- ends in the
// poll budget: NaN output + HEXGNN_ETIMEOUT, never a hang and never a plausible result.
static std::atomic<int> g_reserved_cus{0};
static std::atomic<int> g_persist_override{-1};          // -1: HEXGNN_NO_PERSIST decides, 0: per-layer launches, 1: one launch
static std::mutex g_inflight_mu;
static hipEvent_t g_inflight_ev = nullptr;
static hipStream_t g_inflight_stream = nullptr;
static bool g_inflight_valid = false;
static bool stream_capturing(hipStream_t st) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return true; }
    return cs != hipStreamCaptureStatusNone;
}
static bool other_stream_in_flight(hipStream_t st) {
    std::lock_guard<std::mutex> lock(g_inflight_mu);
    if (!g_inflight_valid || g_inflight_stream == st) return false;
    if (stream_capturing(st)) return false;               // (no event query inside a capture; captured steps are stream-ordered)
    const hipError_t e = hipEventQuery(g_inflight_ev);
    if (e == hipSuccess) { g_inflight_valid = false; return false; }
    (void)hipGetLastError();
    return true;
}
static void note_stack_launch(hipStream_t st) {
    if (stream_capturing(st)) return;
    std::lock_guard<std::mutex> lock(g_inflight_mu);
    if (!g_inflight_ev && hipEventCreateWithFlags(&g_inflight_ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return; }
    if (hipEventRecord(g_inflight_ev, st) == hipSuccess) { g_inflight_stream = st; g_inflight_valid = true; }
    else (void)hipGetLastError();
}
static bool cu_mask_in_force() {
    static const bool m = [] {
        for (const char* k : {"HSA_CU_MASK", "ROC_GLOBAL_CU_MASK", "HSA_CU_MASK_SKIP_INIT"}) { const char* v = getenv(k); if (v && v[0]) return true; }
        return false;
    }();
    return m;
}
// test aid: HEXGNN_STACK_SKEW=<seed> (or hexgnn_debug_stack_skew) delays the blocks unevenly, a new pattern per launch
static std::atomic<unsigned> g_stack_skew_seed{[] { const char* v = getenv("HEXGNN_STACK_SKEW"); return v ? (unsigned)atoi(v) : 0u; }()};
static unsigned stack_skew() {
    static std::atomic<unsigned> counter{0};
    const unsigned seed = g_stack_skew_seed.load();
    if ((seed >> 24) == 0xDEu) return seed;          // "mute block (seed & 0xffff)": the timeout test
    return seed ? (seed + 7919u * counter.fetch_add(1)) & 0x00ffffffu : 0u;
}
template <int NT> static int stack_blocks_per_cu(bool bwd) {
    static int occ[2] = {-1, -1};
    int& o = occ[bwd ? 1 : 0];
    if (o < 0) {
        int nb = 0;
        const size_t lds = 2 * NT * NT * 1024 + RowsLds<NT>::bytes + 64 + 1024;
        const void* f = bwd ? reinterpret_cast<const void*>(&sage_stack_bwd_kernel<NT>) : reinterpret_cast<const void*>(&sage_stack_fwd_kernel<NT>);
        (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, f, 512, lds) != hipSuccess) { (void)hipGetLastError(); nb = 0; }
        o = nb > 1 ? 1 : nb;          // (the protocol budgets one workgroup per CU: 157 KB of LDS at the widths with a row copy)
    }
    return o;
}
static int stack_occupancy(int nt, bool bwd) {
    switch (nt) {
        case 3: return stack_blocks_per_cu<3>(bwd); case 4: return stack_blocks_per_cu<4>(bwd); case 5: return stack_blocks_per_cu<5>(bwd);
        case 6: return stack_blocks_per_cu<6>(bwd); case 7: return stack_blocks_per_cu<7>(bwd); case 8: return stack_blocks_per_cu<8>(bwd);
        default: return 0;
    }
}
static bool persist_fits(int n, int nblocks, int nt, int layers, hipStream_t st, bool bwd) {
    const int ov = g_persist_override.load();
    if (ov == 0) return false;
    if (!(nt >= 3 && layers >= 2 && n > 0 && persist_ready(st))) return false;
    const int blocks = nblocks > 0 ? nblocks : (n + 127) / 128;
    if (blocks > kStackFlagWords || cu_mask_in_force()) return false;
    const bool capturing = stream_capturing(st);
    const int occ = capturing ? 1 : stack_occupancy(nt, bwd);     // (no occupancy query inside a capture: queried by the warm-up)
    if (blocks > occ * g_cu_count - g_reserved_cus.load()) return false;
    return !other_stream_in_flight(st);
}

template <int NT>
static void launch_stack_fwd(StackKArgs a, hipStream_t st) {
    static bool once = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sage_stack_fwd_kernel<NT>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 2 * NT * NT * 1024 + RowsLds<NT>::bytes + 64 + 1024);
        return true;
    }();
    (void)once;
    a.skew = stack_skew();
    {
        KernelTimer kt(HEXGNN_K_SAGE_FWD, st);
        if constexpr (NT >= 3)
            sage_stack_fwd_kernel<NT><<<a.bstart ? a.nblocks : (a.n + 127) / 128, 512, 2 * NT * NT * 1024 + RowsLds<NT>::bytes + 64 + 1024, st>>>(a);
    }
    note_stack_launch(st);
}
template <int NT>
static void launch_stack_bwd(StackKArgs a, hipStream_t st) {
    static bool once = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sage_stack_bwd_kernel<NT>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 2 * NT * NT * 1024 + RowsLds<NT>::bytes + 64 + 1024);
        return true;
    }();
    (void)once;
    a.skew = stack_skew();
    {
        KernelTimer kt(HEXGNN_K_SAGE_BWD, st);
        if constexpr (NT >= 3)
            sage_stack_bwd_kernel<NT><<<a.bstart ? a.nblocks : (a.n + 127) / 128, 512, 2 * NT * NT * 1024 + RowsLds<NT>::bytes + 64 + 1024, st>>>(a);
    }
    note_stack_launch(st);
}

template <int NT>
static void launch_dw(const DwArgs& a, int layers, float* part, hipStream_t st) {
    KernelTimer kt(HEXGNN_K_SAGE_DW, st);
    sage_dw_kernel<NT><<<dim3(a.S, layers), 64 * DwShape<NT>::kWaves, 0, st>>>(a, part);
}

template <int NT>
static void launch_dw16(const Dw16Args& a, int layers, float* part, hipStream_t st) {
    KernelTimer kt(HEXGNN_K_SAGE_DW, st);
    sage_dw16_kernel<NT><<<dim3(a.S, layers), 64 * NT, 0, st>>>(a, part);
}

#define HEXGNN_NT_SWITCH(nt, CALL)                 \
    switch (nt) {                                  \
        case 1: { constexpr int NT_ = 1; CALL; } break; \
        case 2: { constexpr int NT_ = 2; CALL; } break; \
        case 3: { constexpr int NT_ = 3; CALL; } break; \
        case 4: { constexpr int NT_ = 4; CALL; } break; \
        case 5: { constexpr int NT_ = 5; CALL; } break; \
        case 6: { constexpr int NT_ = 6; CALL; } break; \
        case 7: { constexpr int NT_ = 7; CALL; } break; \
        case 8: { constexpr int NT_ = 8; CALL; } break; \
        default: return HEXGNN_EUNSUPPORTED;       \
    }

// Row slices per layer of the batched weight-gradient GEMM.  Exact fp32 (MFMA-bound; two 8-wave workgroups are resident
// per CU): slices of at most ~1024 rows, their number chosen so that (slices x hidden layers) fills a whole number of
// rounds of the 512 resident workgroups -- otherwise the CUs that draw a workgroup of the last, partial round set the
// kernel time (a 256-graph batch of mid-game boards, N = 19 938: 20 x 16 = 320 workgroups took as long as the 496 of the
// start-position batch; 32 x 16 = 512 do not).  Slices stay >= 256 rows.  Split f16 (HBM-bound): (slices x hidden layers)
// fills the 256 CUs in ONE round and halves the slab traffic of the reduce.  The plan sizes its workspace for the larger.
static int dw_slices_fp32(int n, int hidden_layers, bool wide) {
    const int nh = hidden_layers > 0 ? hidden_layers : 1;
    int base = (n + 1023) / 1024;
    if (base < 1) base = 1;
    constexpr int kSlots = 512;
    const int rounds = (base * nh + kSlots - 1) / kSlots;
    int s = rounds * kSlots / nh;
    if (s > n / 256) s = n / 256;
    if (s < base) s = base;
    // one- and two-layer stacks (the per-layer calls of --norm=True, the head stack) get twice the slices: 64 workgroups of a
    // single-layer launch left 7/8 of the 512 slots empty (60 us per layer against 15 us per layer in the batched launch)
    const int cap = (wide && nh <= 2) ? 2 * kDwMaxSlices : kDwMaxSlices;     // (wide: the whole stack has <= 2 hidden layers)
    if (s > cap) s = cap;
    return s;
}
int dw_slices_for(int n, int hidden_layers, int math, int stack_hidden_layers) {
    const int s0 = dw_slices_fp32(n, hidden_layers, stack_hidden_layers <= 2);
    if (math != 1) return s0;
    int s = 256 / (hidden_layers > 0 ? hidden_layers : 1);
    if (s > n / 256) s = n / 256;
    if (s > s0) s = s0;
    if (s < 1) s = 1;
    return s;
}
static int dw_rows_per_slice(int n, int S) {
    int r = (n + S - 1) / S;
    return (r + 31) / 32 * 32;
}

void make_bwd_plan(int n, const StackPlan& p, BwdPlan* b) {
    const size_t slab = align_up(sizeof(float) * (size_t)n * p.hp, 256);
    size_t off = 0;
    b->g_off = off; off += slab * p.L;
    b->S = dw_slices_fp32(n, p.L - (p.small_first ? 1 : 0), p.L - (p.small_first ? 1 : 0) <= 2);
    b->rps = dw_rows_per_slice(n, b->S);
    b->part_off = off; off += align_up(sizeof(float) * dw_slab_count(p.L) * p.hp * (2 * p.hp + 1), 256);
    b->rps0 = 64;             // (128-row slices: 178 workgroups on MIX, 17.2 + 4.3 us with the reduce; 64: 11.4 + 5.3; 32: 9.8 + 7.9)
    b->S0 = (n + b->rps0 - 1) / b->rps0; if (b->S0 < 1) b->S0 = 1;
    b->part0_off = off; off += align_up(sizeof(float) * (size_t)b->S0 * p.hp * 17, 256);
    b->total = off;
}

int fill_pack_args(const StackPlan& p, int c_in, int hidden, const float* const* wl, const float* const* bl,
                   const float* const* wr, PackArgs* pa) {
    for (int l = 0; l < p.L; ++l) {
        if (!wl[l] || !bl[l] || !wr[l]) return HEXGNN_EINVAL;
        pa->p.wl[l] = wl[l]; pa->p.bl[l] = bl[l]; pa->p.wr[l] = wr[l];
        pa->fwd_off[l] = p.fwd_off[l]; pa->bwd_off[l] = p.bwd_off[l]; pa->bias_off[l] = p.bias_off[l];
    }
    pa->hp = p.hp; pa->nt = p.nt; pa->L = p.L; pa->c_in = c_in; pa->hidden = hidden; pa->small_first = p.small_first;
    pa->flag_off = p.flag_off;
    return HEXGNN_OK;
}

int launch_pack(const StackPlan& p, int c_in, int hidden, const float* const* wl, const float* const* bl,
                const float* const* wr, void* wpack, hipStream_t st, int math, unsigned* zero_maxima) {
    if (!wl && !bl && !wr && math == 0) return HEXGNN_OK;      // packed already (hexgnn_csr_build_grouped_pack of this forward)
    if (!wl || !bl || !wr) return HEXGNN_EINVAL;
    PackArgs pa;
    const int rcp = fill_pack_args(p, c_in, hidden, wl, bl, wr, &pa);
    if (rcp != HEXGNN_OK) return rcp;
    const int pack_elems = 2 * p.nt * p.nt * 256;
    sage_pack_kernel<<<dim3((pack_elems + 255) / 256, p.L), 256, 0, st>>>(pa, (char*)wpack);
    if (math == 1) {   // overwrite the hidden layers' weight packs with the split-f16 layout (biases / raw layer stay fp32)
        Pack16Args pb;
        pb.p = pa.p;
        for (int l = 0; l < p.L; ++l) { pb.fwd_off[l] = p.fwd_off[l]; pb.bwd_off[l] = p.bwd_off[l]; pb.bias_off[l] = p.bias_off[l]; }
        pb.nt = p.nt; pb.L = p.L; pb.hidden = hidden; pb.first_hidden = p.small_first ? 1 : 0; pb.hp = p.hp;
        pb.zero_maxima = zero_maxima;
        const int nh = p.L - pb.first_hidden;
        const int elems = 2 * 2 * p.nt * p.nt * 256;
        if (nh > 0) {
            sage_wscale_kernel<<<nh, 1024, 0, st>>>(pb, (char*)wpack);
            sage_pack16_kernel<<<dim3((elems + 255) / 256, nh), 256, 0, st>>>(pb, (char*)wpack);
        }
    }
    return HEXGNN_OK;
}

int launch_weight_grads(int n, int c_in, int hidden, const StackPlan& p, const BwdPlan& b, const float* x,
                        int x_stride, const float* acts, const char* sv, const float* G, float* const* d_wl,
                        float* const* d_bl, float* const* d_wr, float* part, float* part0, hipStream_t st,
                        int math, const unsigned* xmax, const unsigned* gmax, bool hidden_only_no_reduce,
                        int layer_lo, int layer_hi) {
    const size_t slab = (size_t)n * p.hp;
    const int lo = layer_lo < 0 ? (p.small_first ? 1 : 0) : layer_lo;
    const int first_hidden = lo;                    // first layer of this launch (hidden-input layers only)
    const int nh = (layer_hi < 0 ? p.L : layer_hi) - lo;
    bool first_done = false;
    if (nh > 0) {
        DwArgs da;
        DwReduceArgs ra;
        for (int i = 0; i < nh; ++i) {
            const int l = first_hidden + i;
            da.xin[i] = l == 0 ? x : acts + slab * (l - 1);
            da.agg[i] = (const float*)(sv + p.agg_off[l]);
            da.g[i] = G + slab * l;
            ra.dwl[i] = d_wl[l]; ra.dbl[i] = d_bl[l]; ra.dwr[i] = d_wr[l];
        }
        const int S = dw_slices_for(n, nh, (math == 1 && xmax && gmax) ? 1 : 0, p.L - (p.small_first ? 1 : 0));
        const int rps = dw_rows_per_slice(n, S);
        da.n = n; da.rows_per_slice = rps; da.S = S;
        ra.S = S; ra.hp = p.hp; ra.hidden = hidden;
        if (math == 1 && xmax && gmax) {
            Dw16Args d16;
            for (int i = 0; i < nh; ++i) { d16.xin[i] = da.xin[i]; d16.agg[i] = da.agg[i]; d16.g[i] = da.g[i]; }
            d16.xmax = xmax; d16.gmax = gmax; d16.first_hidden = first_hidden;
            d16.n = n; d16.rows_per_slice = rps; d16.S = S;
            HEXGNN_NT_SWITCH(p.nt, (launch_dw16<NT_>(d16, nh, part, st)));
        } else {
            HEXGNN_NT_SWITCH(p.nt, (launch_dw<NT_>(da, nh, part, st)));
        }
        const int tot = hidden * (2 * hidden + 1);
        if (!hidden_only_no_reduce) {
            if (p.small_first && layer_lo < 0) {       // + the raw first layer's row-slice partials in the same launch
                const int nrb = (tot + 255) / 256;
                sage_dw_reduce_first_kernel<<<nrb * nh + b.S0, 256, 0, st>>>(ra, part, nrb, nh, n, c_in, b.rps0, G,
                                                                            (const float*)(sv + p.agg_off[0]), x, x_stride, part0);
                first_done = true;
            } else {
                sage_dw_reduce_kernel<<<dim3((tot + 255) / 256, nh), 256, 0, st>>>(ra, part);
            }
        }
    }
    if (p.small_first && !hidden_only_no_reduce && layer_lo < 0) {
        if (!first_done)
            sage_first_dw_kernel<<<b.S0, 256, 0, st>>>(n, c_in, p.hp, b.rps0, G, (const float*)(sv + p.agg_off[0]), x,
                                                       x_stride, part0);
        const int tot = hidden * (2 * c_in + 1);
        sage_first_dw_reduce_kernel<<<tot, 64, 0, st>>>(b.S0, p.hp, hidden, c_in, part0, d_wl[0], d_bl[0], d_wr[0]);
    }
    return HEXGNN_OK;
}

}  // namespace hexgnn

using namespace hexgnn;

extern "C" {

size_t hexgnn_sage_stack_pack_bytes(int c_in, int hidden, int num_layers) {
    if (hidden > 16 * kMaxNT) { WidePlan w; return wide_make_plan(0, c_in, hidden, num_layers, &w) == HEXGNN_OK ? w.pack_bytes : 0; }
    StackPlan p;
    if (make_plan(0, c_in, hidden, num_layers, &p) != HEXGNN_OK) return 0;
    return p.pack_bytes;
}

size_t hexgnn_sage_stack_saved_bytes(int n, int c_in, int hidden, int num_layers) {
    if (hidden > 16 * kMaxNT) { WidePlan w; return (n >= 0 && wide_make_plan(n, c_in, hidden, num_layers, &w) == HEXGNN_OK) ? w.saved_bytes : 0; }
    StackPlan p;
    if (n < 0 || make_plan(n, c_in, hidden, num_layers, &p) != HEXGNN_OK) return 0;
    return p.saved_bytes;
}

int hexgnn_sage_stack_forward(int n, int c_in, int hidden, int num_layers, const int* rowptr, const int* col,
                              const float* invdeg, const float* x, int x_stride, const float* const* wl,
                              const float* const* bl, const float* const* wr, void* wpack, float* acts,
                              void* saved, int need_backward, int flags, hexgnn_stream_t stream_) {
    return hexgnn_sage_stack_forward_blocks(n, c_in, hidden, num_layers, rowptr, col, invdeg, x, x_stride, wl, bl, wr, wpack, acts,
                                            saved, need_backward, flags, nullptr, 0, stream_);
}

// a block table is usable when it can be a partition of [0, n) into at most kStackFlagWords pieces of at most 128 rows (its
// CONTENT is device data: checked by the kernel, block by block)
static bool block_table_ok(int n, const int* block_starts, int num_blocks) {
    if (!block_starts) return num_blocks == 0;
    return num_blocks >= (n + 127) / 128 && num_blocks >= 1 && num_blocks <= kStackFlagWords && num_blocks <= n;
}

int hexgnn_sage_stack_forward_blocks(int n, int c_in, int hidden, int num_layers, const int* rowptr, const int* col,
                                     const float* invdeg, const float* x, int x_stride, const float* const* wl,
                                     const float* const* bl, const float* const* wr, void* wpack, float* acts,
                                     void* saved, int need_backward, int flags, const int* block_starts, int num_blocks,
                                     hexgnn_stream_t stream_) {
    hipStream_t st = (hipStream_t)stream_;
    StackPlan p;
    if (n < 0 || (flags & ~HEXGNN_SAGE_LINEAR_LAST)) return HEXGNN_EINVAL;
    if (n > 0 && !block_table_ok(n, block_starts, num_blocks)) return HEXGNN_EINVAL;
    if (hidden > 16 * kMaxNT) {        // 129..256: the plain kernels of wide.hip (always materialise the aggregates in `saved`)
        if (!wl || !bl || !wr || !wpack) return HEXGNN_EINVAL;
        if (n > 0 && (!rowptr || !col || !invdeg || !x || !acts)) return HEXGNN_EINVAL;
        return wide_stack_forward(n, c_in, hidden, num_layers, rowptr, col, invdeg, x, x_stride, wl, bl, wr, wpack, acts, saved,
                                  flags, st);
    }
    int rc = make_plan(n, c_in, hidden, num_layers, &p);
    if (rc != HEXGNN_OK) return rc;
    if (!wpack || ((!wl || !bl || !wr) && (wl || bl || wr))) return HEXGNN_EINVAL;
    if (n > 0 && (!rowptr || !col || !invdeg || !x || !acts)) return HEXGNN_EINVAL;
    if (need_backward && !saved) return HEXGNN_EINVAL;
    if (p.small_first ? x_stride < c_in : x_stride != p.hp) return HEXGNN_EINVAL;

    rc = launch_pack(p, c_in, hidden, wl, bl, wr, wpack, st, 0);     // (all three arrays NULL: packed by the CSR call already)
    if (rc != HEXGNN_OK) return rc;
    if (n == 0) return check_launch();

    const size_t slab = (size_t)n * p.hp;
    char* wp = (char*)wpack;
    char* sv = (char*)saved;
    rc = take_stack_status();
    if (rc != HEXGNN_OK) return rc;
    const int fh = p.small_first ? 1 : 0;
    bool one_launch = persist_fits(n, block_starts ? num_blocks : 0, p.nt, p.L - fh, st, false);
    if (!one_launch && block_starts && persist_fits(n, 0, p.nt, p.L - fh, st, false)) {
        one_launch = true;           // the table's blocks do not fit the resident-workgroup budget, the default 128-row blocks do
        block_starts = nullptr; num_blocks = 0;
    }
    for (int l = 0; l < p.L; ++l) {
        float* y = acts + slab * l;
        const float* bias = (const float*)(wp + p.bias_off[l]);
        float* agg = need_backward ? (float*)(sv + p.agg_off[l]) : nullptr;
        const int relu = !(l == p.L - 1 && (flags & HEXGNN_SAGE_LINEAR_LAST));
        if (l == 0 && p.small_first) {
            KernelTimer kt(HEXGNN_K_SAGE_FIRST, st);
            sage_first_fwd_kernel<<<(n + 31) / 32, 256, 0, st>>>(n, c_in, p.hp, rowptr, col, invdeg, x, x_stride,
                                                              (const float*)(wp + p.fwd_off[0]), bias, y, agg, relu);
        } else if (one_launch) {
            StackKArgs a{};
            a.n = n; a.l_first = l; a.n_layers = p.L - l;
            a.relu_last = !(flags & HEXGNN_SAGE_LINEAR_LAST); a.last_of_stack = p.L - 1; a.tap_layer = -1;
            a.rowptr = rowptr; a.col = col; a.invdeg = invdeg;
            a.in0 = l == 0 ? x : acts + slab * (l - 1);
            a.slabs = acts; a.slab = slab;
            a.w0 = wp + p.fwd_off[l]; a.wstride = p.L - l > 1 ? p.fwd_off[l + 1] - p.fwd_off[l] : 0;
            a.b0 = wp + p.bias_off[l];
            a.agg0 = need_backward ? sv + p.agg_off[l] : nullptr;
            a.astride = p.L - l > 1 ? p.agg_off[l + 1] - p.agg_off[l] : 0;
            a.flags = reinterpret_cast<unsigned*>(wp + p.flag_off);
            a.bstart = block_starts; a.nblocks = num_blocks;
            a.status = g_stack_status;
            HEXGNN_NT_SWITCH(p.nt, (launch_stack_fwd<NT_>(a, st)));
            break;
        } else {
            const float* xin = l == 0 ? x : acts + slab * (l - 1);
            HEXGNN_NT_SWITCH(p.nt, (launch_fwd<NT_>(n, rowptr, col, invdeg, xin, wp + p.fwd_off[l], bias, y, agg, relu, st)));
        }
    }
    return check_launch();
}

size_t hexgnn_sage_stack_backward_workspace_bytes(int n, int c_in, int hidden, int num_layers) {
    if (hidden > 16 * kMaxNT) { WidePlan w; return (n >= 0 && wide_make_plan(n, c_in, hidden, num_layers, &w) == HEXGNN_OK) ? w.bwd_bytes : 0; }
    StackPlan p;
    if (n < 0 || make_plan(n, c_in, hidden, num_layers, &p) != HEXGNN_OK) return 0;
    BwdPlan b;
    make_bwd_plan(n, p, &b);
    return b.total;
}

int hexgnn_sage_stack_backward(int n, int c_in, int hidden, int num_layers, const int* rowptr, const int* col,
                               const int* rowptr_t, const int* col_t, const float* invdeg, const float* x,
                               int x_stride, const float* acts, const void* saved, const void* wpack,
                               const float* dy, float* dx, float* const* d_wl, float* const* d_bl,
                               float* const* d_wr, void* workspace, size_t workspace_bytes, int flags,
                               hexgnn_stream_t stream_) {
    return hexgnn_sage_stack_backward_tap(n, c_in, hidden, num_layers, rowptr, col, rowptr_t, col_t, invdeg, x, x_stride, acts,
                                          saved, wpack, dy, dx, d_wl, d_bl, d_wr, workspace, workspace_bytes, flags, -1,
                                          nullptr, stream_);
}

int hexgnn_sage_stack_backward_tap(int n, int c_in, int hidden, int num_layers, const int* rowptr, const int* col,
                                   const int* rowptr_t, const int* col_t, const float* invdeg, const float* x,
                                   int x_stride, const float* acts, const void* saved, const void* wpack,
                                   const float* dy, float* dx, float* const* d_wl, float* const* d_bl,
                                   float* const* d_wr, void* workspace, size_t workspace_bytes, int flags,
                                   int tap_layer, float* tap_out, hexgnn_stream_t stream_) {
    return hexgnn_sage_stack_backward_blocks(n, c_in, hidden, num_layers, rowptr, col, rowptr_t, col_t, invdeg, x, x_stride, acts,
                                             saved, wpack, dy, dx, d_wl, d_bl, d_wr, workspace, workspace_bytes, flags,
                                             tap_layer, tap_out, nullptr, 0, stream_);
}

int hexgnn_sage_stack_backward_blocks(int n, int c_in, int hidden, int num_layers, const int* rowptr, const int* col,
                                      const int* rowptr_t, const int* col_t, const float* invdeg, const float* x,
                                      int x_stride, const float* acts, const void* saved, const void* wpack,
                                      const float* dy, float* dx, float* const* d_wl, float* const* d_bl,
                                      float* const* d_wr, void* workspace, size_t workspace_bytes, int flags,
                                      int tap_layer, float* tap_out, const int* block_starts, int num_blocks,
                                      hexgnn_stream_t stream_) {
    (void)rowptr; (void)col;
    hipStream_t st = (hipStream_t)stream_;
    StackPlan p;
    if (n < 0 || (flags & ~(HEXGNN_SAGE_LINEAR_LAST | HEXGNN_SAGE_DY_IN_PLACE))) return HEXGNN_EINVAL;
    if (n > 0 && !block_table_ok(n, block_starts, num_blocks)) return HEXGNN_EINVAL;
    if ((flags & HEXGNN_SAGE_DY_IN_PLACE) && (flags & HEXGNN_SAGE_LINEAR_LAST)) return HEXGNN_EINVAL;
    if (hidden > 16 * kMaxNT) {
        if (!d_wl || !d_bl || !d_wr || !wpack || !saved) return HEXGNN_EINVAL;
        if (n > 0 && (!rowptr_t || !col_t || !invdeg || !x || !acts || !dy)) return HEXGNN_EINVAL;
        return wide_stack_backward(n, c_in, hidden, num_layers, rowptr_t, col_t, invdeg, x, x_stride, acts, saved, wpack, dy, dx,
                                   d_wl, d_bl, d_wr, workspace, workspace_bytes, flags, tap_layer, tap_out, st);
    }
    int rc = make_plan(n, c_in, hidden, num_layers, &p);
    if (rc != HEXGNN_OK) return rc;
    BwdPlan b;
    make_bwd_plan(n, p, &b);
    if (!workspace || workspace_bytes < b.total) return HEXGNN_EWORKSPACE;
    if (!d_wl || !d_bl || !d_wr || !wpack || !saved) return HEXGNN_EINVAL;
    for (int l = 0; l < p.L; ++l) if (!d_wl[l] || !d_bl[l] || !d_wr[l]) return HEXGNN_EINVAL;
    if (n > 0 && (!rowptr_t || !col_t || !invdeg || !x || !acts || !dy)) return HEXGNN_EINVAL;

    const size_t slab = (size_t)n * p.hp;
    char* ws = (char*)workspace;
    const char* wp = (const char*)wpack;
    const char* sv = (const char*)saved;
    float* G = (float*)(ws + b.g_off);
    float* part = (float*)(ws + b.part_off);
    float* part0 = (float*)(ws + b.part0_off);

    if (n == 0) {  // empty batch: all parameter gradients are zero
        for (int l = 0; l < p.L; ++l) {
            const int in = (l == 0) ? c_in : hidden;
            (void)hipMemsetAsync(d_wl[l], 0, sizeof(float) * (size_t)hidden * in, st);
            (void)hipMemsetAsync(d_wr[l], 0, sizeof(float) * (size_t)hidden * in, st);
            (void)hipMemsetAsync(d_bl[l], 0, sizeof(float) * (size_t)hidden, st);
        }
        return check_launch();
    }

    // data-gradient chain, top layer first: G_{L-1} = dy * [y_{L-1} > 0], then per hidden-input layer l one launch
    //   G_{l-1} = ( [ sum_{T} G_l / deg | G_l ] [W_l ; W_r] ) * [y_{l-1} > 0]      (l == 0: the stack-input gradient dx, unmasked)
    const int first_hidden = p.small_first ? 1 : 0;
    const int q4 = p.hp / 4;
    const unsigned cgrid = (unsigned)(((int64_t)n * q4 + 255) / 256);
    if (flags & HEXGNN_SAGE_DY_IN_PLACE) {
        // the caller's producer (hexgnn_head_backward with HEXGNN_HEAD_MASK_DH) wrote G_{L-1} = dy * [y_{L-1} > 0] straight
        // into its slab of the workspace: nothing to combine
        if (dy != G + slab * (p.L - 1)) return HEXGNN_EINVAL;
    } else {
        KernelTimer kt(HEXGNN_K_COMBINE, st);
        const bool relu_top = !(flags & HEXGNN_SAGE_LINEAR_LAST);
        sage_combine_kernel<<<cgrid, 256, 0, st>>>(n, p.hp, rowptr_t, col_t, dy, nullptr, relu_top ? acts + slab * (p.L - 1) : nullptr,
                                                   G + slab * (p.L - 1));
    }
    if (tap_out && (tap_layer < 0 || tap_layer >= p.L - 1)) return HEXGNN_EINVAL;
    rc = take_stack_status();
    if (rc != HEXGNN_OK) return rc;
    const int lo = (first_hidden == 0 && !dx) ? 1 : first_hidden;        // last layer whose data gradient is wanted
    bool one_launch = persist_fits(n, block_starts ? num_blocks : 0, p.nt, p.L - lo, st, true);
    if (!one_launch && block_starts && persist_fits(n, 0, p.nt, p.L - lo, st, true)) {
        one_launch = true;           // (as in the forward call)
        block_starts = nullptr; num_blocks = 0;
    }
    if (one_launch) {
        StackKArgs a{};
        a.n = n; a.l_first = p.L - 1; a.n_layers = p.L - lo;
        a.tap_layer = tap_out ? tap_layer : -1; a.tap_out = tap_out;
        a.rowptr = rowptr_t; a.col = col_t; a.invdeg = invdeg;
        a.in0 = G + slab * (p.L - 1);
        a.slabs = G; a.slab = slab; a.masks = acts; a.dx = dx;
        a.w0 = wp + p.bwd_off[p.L - 1]; a.wstride = p.bwd_off[p.L - 1] - p.bwd_off[p.L - 2];
        a.flags = reinterpret_cast<unsigned*>(const_cast<char*>(wp) + p.flag_off) + kStackFlagWords;
        a.bstart = block_starts; a.nblocks = num_blocks;
        a.status = g_stack_status;
        HEXGNN_NT_SWITCH(p.nt, (launch_stack_bwd<NT_>(a, st)));
    }
    for (int l = p.L - 1; l >= first_hidden && !one_launch; --l) {
        float* out = l >= 1 ? G + slab * (l - 1) : dx;
        if (!out) break;                                   // l == 0 and nobody asked for the input gradient
        const float* ymask = l >= 1 ? acts + slab * (l - 1) : nullptr;
        // (the gradient w.r.t. layer tap_layer's OUTPUT, before its ReLU mask, leaves the same launch: the epilogue stores
        // the rows twice)
        float* tap = (tap_out && l - 1 == tap_layer) ? tap_out : nullptr;
        HEXGNN_NT_SWITCH(p.nt, (launch_bwd<NT_>(n, rowptr_t, col_t, invdeg, G + slab * l, wp + p.bwd_off[l], ymask, out, st, tap)));
    }

    rc = launch_weight_grads(n, c_in, hidden, p, b, x, x_stride, acts, sv, G, d_wl, d_bl, d_wr, part, part0, st);
    if (rc != HEXGNN_OK) return rc;
    return check_launch();
}

/* ---- SAGE stack with the whole-batch LayerNorm of --norm=True between every contraction and its ReLU ------------------------
 * (CachifiedGNN.forward with norms, GN0/models.py:261-294: conv -> norm -> relu per layer).  One call per direction instead of
 * a SAGE call + a norm call per layer: the weights are packed once, and the weight gradients of all layers are ONE batched
 * GEMM + one reduce (per-layer launches of 64-128 workgroups ran at a quarter of the batched rate). */
size_t hexgnn_sage_norm_stack_backward_workspace_bytes(int n, int c_in, int hidden, int num_layers) {
    StackPlan p;
    if (n < 0 || make_plan(n, c_in, hidden, num_layers, &p) != HEXGNN_OK) return 0;
    BwdPlan b;
    make_bwd_plan(n, p, &b);
    return b.total + align_up(sizeof(float) * (size_t)n * p.hp, 256);
}

int hexgnn_sage_norm_stack_forward(int n, int c_in, int hidden, int num_layers, const int* rowptr, const int* col,
                                   const float* invdeg, const float* x, int x_stride, const float* const* wl,
                                   const float* const* bl, const float* const* wr, const float* const* nw,
                                   const float* const* nb, float eps, void* wpack, float* pre, float* acts, void* saved,
                                   float* stats, void* norm_ws, size_t norm_ws_bytes, int need_backward,
                                   hexgnn_stream_t stream_) {
    return hexgnn_sage_norm_stack_forward_live(n, nullptr, c_in, hidden, num_layers, rowptr, col, invdeg, x, x_stride, wl, bl,
                                               wr, nw, nb, eps, wpack, pre, acts, saved, stats, norm_ws, norm_ws_bytes,
                                               need_backward, stream_);
}

int hexgnn_sage_norm_stack_forward_live(int n, const int* n_live, int c_in, int hidden, int num_layers, const int* rowptr,
                                        const int* col, const float* invdeg, const float* x, int x_stride,
                                        const float* const* wl, const float* const* bl, const float* const* wr,
                                        const float* const* nw, const float* const* nb, float eps, void* wpack, float* pre,
                                        float* acts, void* saved, float* stats, void* norm_ws, size_t norm_ws_bytes,
                                        int need_backward, hexgnn_stream_t stream_) {
    hipStream_t st = (hipStream_t)stream_;
    StackPlan p;
    if (n < 0) return HEXGNN_EINVAL;
    if (n_live && need_backward) return HEXGNN_EINVAL;       // a live row count is an acting-time (forward-only) notion
    int rc = make_plan(n, c_in, hidden, num_layers, &p);
    if (rc != HEXGNN_OK) return rc;
    if (!wl || !bl || !wr || !nw || !nb || !wpack || !stats || !norm_ws) return HEXGNN_EINVAL;
    if (n > 0 && (!rowptr || !col || !invdeg || !x || !acts || !pre)) return HEXGNN_EINVAL;
    if (need_backward && !saved) return HEXGNN_EINVAL;
    if (p.small_first ? x_stride < c_in : x_stride != p.hp) return HEXGNN_EINVAL;
    for (int l = 0; l < p.L; ++l) if (!nw[l] || !nb[l]) return HEXGNN_EINVAL;
    rc = launch_pack(p, c_in, hidden, wl, bl, wr, wpack, st, 0);
    if (rc != HEXGNN_OK) return rc;
    if (n == 0) return check_launch();
    const size_t slab = (size_t)n * p.hp;
    char* wp = (char*)wpack;
    char* sv = (char*)saved;
    for (int l = 0; l < p.L; ++l) {
        float* y = pre + slab * l;
        const float* bias = (const float*)(wp + p.bias_off[l]);
        float* agg = need_backward ? (float*)(sv + p.agg_off[l]) : nullptr;
        if (l == 0 && p.small_first) {
            KernelTimer kt(HEXGNN_K_SAGE_FIRST, st);
            sage_first_fwd_kernel<<<(n + 31) / 32, 256, 0, st>>>(n, c_in, p.hp, rowptr, col, invdeg, x, x_stride,
                                                              (const float*)(wp + p.fwd_off[0]), bias, y, agg, 0);
        } else {
            const float* xin = l == 0 ? x : acts + slab * (l - 1);
            HEXGNN_NT_SWITCH(p.nt, (launch_fwd<NT_>(n, rowptr, col, invdeg, xin, wp + p.fwd_off[l], bias, y, agg, 0, st)));
        }
        rc = hexgnn_graph_layernorm_forward_live(n, n_live, hidden, y, nw[l], nb[l], eps, 1, acts + slab * l, stats + 2 * l,
                                                 norm_ws, norm_ws_bytes, stream_);
        if (rc != HEXGNN_OK) return rc;
    }
    return check_launch();
}

int hexgnn_sage_norm_stack_backward(int n, int c_in, int hidden, int num_layers, const int* rowptr_t, const int* col_t,
                                    const float* invdeg, const float* x, int x_stride, const float* pre,
                                    const float* acts, const void* saved, const void* wpack, const float* stats,
                                    const float* const* nw, float eps, const float* dy, float* dx, float* const* d_wl,
                                    float* const* d_bl, float* const* d_wr, float* const* d_nw, float* const* d_nb,
                                    void* workspace, size_t workspace_bytes, void* norm_ws, size_t norm_ws_bytes,
                                    hexgnn_stream_t stream_) {
    hipStream_t st = (hipStream_t)stream_;
    StackPlan p;
    if (n < 0) return HEXGNN_EINVAL;
    int rc = make_plan(n, c_in, hidden, num_layers, &p);
    if (rc != HEXGNN_OK) return rc;
    BwdPlan b;
    make_bwd_plan(n, p, &b);
    const size_t tmp_off = b.total;
    if (!workspace || workspace_bytes < b.total + align_up(sizeof(float) * (size_t)n * p.hp, 256)) return HEXGNN_EWORKSPACE;
    if (!d_wl || !d_bl || !d_wr || !d_nw || !d_nb || !nw || !wpack || !saved || !stats || !norm_ws) return HEXGNN_EINVAL;
    for (int l = 0; l < p.L; ++l)
        if (!d_wl[l] || !d_bl[l] || !d_wr[l] || !d_nw[l] || !d_nb[l] || !nw[l]) return HEXGNN_EINVAL;
    if (n > 0 && (!rowptr_t || !col_t || !invdeg || !x || !acts || !pre || !dy)) return HEXGNN_EINVAL;
    const size_t slab = (size_t)n * p.hp;
    char* ws = (char*)workspace;
    const char* wp = (const char*)wpack;
    const char* sv = (const char*)saved;
    float* G = (float*)(ws + b.g_off);
    float* part = (float*)(ws + b.part_off);
    float* part0 = (float*)(ws + b.part0_off);
    float* tmp = (float*)(ws + tmp_off);
    if (n == 0) {
        for (int l = 0; l < p.L; ++l) {
            const int in = (l == 0) ? c_in : hidden;
            (void)hipMemsetAsync(d_wl[l], 0, sizeof(float) * (size_t)hidden * in, st);
            (void)hipMemsetAsync(d_wr[l], 0, sizeof(float) * (size_t)hidden * in, st);
            (void)hipMemsetAsync(d_bl[l], 0, sizeof(float) * (size_t)hidden, st);
            (void)hipMemsetAsync(d_nw[l], 0, sizeof(float) * (size_t)hidden, st);
            (void)hipMemsetAsync(d_nb[l], 0, sizeof(float) * (size_t)hidden, st);
        }
        return check_launch();
    }
    // top layer first: norm backward (mask by the layer's output, d gamma / d beta) gives G_l = gradient at the contraction's
    // output; the layer kernel turns it into the gradient at the layer's input = the next norm's dy (one scratch slab)
    const int first_hidden = p.small_first ? 1 : 0;
    const float* dcur = dy;
    for (int l = p.L - 1; l >= 0; --l) {
        rc = hexgnn_graph_layernorm_backward(n, hidden, pre + slab * l, acts + slab * l, nw[l], stats + 2 * l, dcur, eps, 1,
                                             G + slab * l, d_nw[l], d_nb[l], norm_ws, norm_ws_bytes, stream_);
        if (rc != HEXGNN_OK) return rc;
        if (l < first_hidden) break;
        float* out = l >= 1 ? tmp : dx;
        if (!out) break;
        HEXGNN_NT_SWITCH(p.nt, (launch_bwd<NT_>(n, rowptr_t, col_t, invdeg, G + slab * l, wp + p.bwd_off[l], nullptr, out, st)));
        dcur = tmp;
    }
    rc = launch_weight_grads(n, c_in, hidden, p, b, x, x_stride, acts, sv, G, d_wl, d_bl, d_wr, part, part0, st);
    if (rc != HEXGNN_OK) return rc;
    return check_launch();
}

int hexgnn_stack_status(int clear) {
    if (!g_stack_status) return HEXGNN_OK;
    const int c = *reinterpret_cast<volatile int*>(g_stack_status);
    if (c != 0 && clear) *g_stack_status = 0;
    return c;
}

int hexgnn_stack_reserve_cus(int cus) {
    if (cus < 0) return HEXGNN_EINVAL;
    return g_reserved_cus.exchange(cus);
}

int hexgnn_stack_block_budget(void) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    if (cu_mask_in_force()) return 0;
    const int b = cus - g_reserved_cus.load();
    return b < 0 ? 0 : (b > kStackFlagWords ? kStackFlagWords : b);
}

int hexgnn_debug_stack_mode(int persist, unsigned skew_seed) {
    if (persist < -1 || persist > 1) return HEXGNN_EINVAL;
    g_persist_override.store(persist);
    g_stack_skew_seed.store(skew_seed);
    return HEXGNN_OK;
}

namespace hexgnn {
// test aid: `blocks` workgroups of 1024 threads that keep their CUs' memory pipes busy for ~usec microseconds (a streaming
// kernel beside the one-launch stack kernels: uneven load for the hand-over's stress test)
__global__ __launch_bounds__(1024) void debug_occupy_kernel(const f32x4* __restrict__ src, size_t words4, f32x4* __restrict__ sink,
                                                            unsigned long long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    size_t i = ((size_t)blockIdx.x * 1024 + threadIdx.x) % words4;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { acc += src[i]; i += 1024 * 61; if (i >= words4) i -= words4; }
    }
    if (acc[0] == 1.2345e-30f) sink[0] = acc;      // (never true: keeps the loads)
}
}  // namespace hexgnn
int hexgnn_debug_occupy(int blocks, int usec, const void* buffer, size_t buffer_bytes, void* sink, hexgnn_stream_t stream_) {
    if (blocks <= 0 || usec <= 0 || !buffer || buffer_bytes < 16 * 1024 * 64 || !sink) return HEXGNN_EINVAL;
    hexgnn::debug_occupy_kernel<<<blocks, 1024, 0, (hipStream_t)stream_>>>((const f32x4*)buffer, buffer_bytes / 16, (f32x4*)sink,
                                                                          (unsigned long long)usec * 100ull);   // 100 MHz clock
    return check_launch();
}

#ifdef HEXGNN_STAMPS
int hexgnn_debug_stamp_block(int block) {
    if (hipDeviceSynchronize() != hipSuccess) return HEXGNN_EHIP;
    return hipMemcpyToSymbol(HIP_SYMBOL(hexgnn::g_stamp_block), &block, sizeof(int)) == hipSuccess ? HEXGNN_OK : HEXGNN_EHIP;
}
// profiling builds only: s_memtime stamps of the layer-major kernels' last launches -> `out` (host pointer, 2 x 8 x 8)
int hexgnn_debug_layer_stamps(unsigned long long* out, int capacity) {
    if (capacity < 128) return HEXGNN_EINVAL;
    if (hipDeviceSynchronize() != hipSuccess) return HEXGNN_EHIP;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(hexgnn::g_lstamps), sizeof(unsigned long long) * 128) != hipSuccess) return HEXGNN_EHIP;
    if (capacity >= 384) {       // + the one-launch stack kernels' stamps, [2][16][8]
        if (hipMemcpyFromSymbol(out + 128, HIP_SYMBOL(hexgnn::g_pstamps), sizeof(unsigned long long) * 256) != hipSuccess) return HEXGNN_EHIP;
        return 384;
    }
    return 128;
}
#endif
}  // extern "C"
