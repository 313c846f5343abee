// hidden_channels 129..256: the SAGE stack and the head tail beyond the widths the LDS-resident kernels are compiled for.
//
// Reference: CachifiedGNN.grow_width / DuellingTwoHeaded.grow_width widen a model to ANY width (GN0/models.py:187-238,
// 497-508); no configuration of the reference goes past 110, so this path is built for coverage first:
// deterministic kernels -- a mean gather (HBM-bound), an exact-fp32 MFMA GEMM (round 4: weights streamed through LDS in K
// chunks of 16 shared by the eight waves of a 128-row block, row operands prefetched in registers; no fusion with the
// gather), transposed weight copies for the backward, a weight-gradient GEMM over 128 x 128 output tiles per row slice (round 4:
// both operands staged through LDS; it was one wave per 16 x 16 tile reading straight from L2), simple per-graph head-tail kernels.  Same arithmetic as the narrow kernels
// (v_mfma_f32_16x16x4_f32, fmaf chains in k order), same saved-tensor layout (acts [L][n][HP], agg per layer), same C entry
// points (sage.hip / head.hip dispatch here when hidden > 128).  Not covered at these widths: the fused per-graph kernels, the
// norms, the two_headed tail and the HexAra pieces (their entry points refuse loudly).
#include "hexgnn_internal.h"

namespace hexgnn {

int padded_width_wide(int hidden) {
    if (hidden <= 0 || hidden > kWideMaxHidden) return -1;
    return 16 * ((hidden + 15) / 16);
}

// ---- plans -------------------------------------------------------------------------------------------------------------
// wpack (per layer): raw first layer [HP][8] Wl, [HP][8] Wr, bias[HP]; hidden layer: WlP, WrP (forward operands, [HP][HP],
// row = output channel, zero padded), WlT, WrT (their transposes: backward operands), bias[HP].
int wide_make_plan(int n, int c_in, int hidden, int L, WidePlan* p) {
    const int hp = padded_width_wide(hidden);
    if (hp < 0 || L < 1 || L > kMaxLayers) return HEXGNN_EUNSUPPORTED;
    if (c_in != hidden && (c_in < 1 || c_in > kSmallCin)) return HEXGNN_EUNSUPPORTED;
    if (n > 0 && (size_t)n * hp * sizeof(float) > 0x7fffffffull) return HEXGNN_EUNSUPPORTED;
    p->hp = hp; p->L = L; p->small_first = c_in != hidden;
    size_t off = 0, soff = 0;
    const size_t sq = sizeof(float) * (size_t)hp * hp;
    for (int l = 0; l < L; ++l) {
        if (l == 0 && p->small_first) {
            p->w_off[l] = off; off += sizeof(float) * (size_t)hp * kSmallCin * 2;
            p->agg_off[l] = soff; soff += align_up(sizeof(float) * (size_t)n * kSmallCin, 256);
        } else {
            p->w_off[l] = off; off += 4 * sq;
            p->agg_off[l] = soff; soff += align_up(sizeof(float) * (size_t)n * hp, 256);
        }
        p->bias_off[l] = off; off += align_up(sizeof(float) * hp, 256);
    }
    p->pack_bytes = off;
    p->saved_bytes = soff;
    // backward workspace: G [L][n][HP], two scratch slabs (dAgg, dXs)
    const size_t slab = align_up(sizeof(float) * (size_t)n * hp, 256);
    p->g_off = 0;
    p->tmp_off = slab * L;
    p->part_off = slab * (L + 2);               // row-slice partials of ONE weight-gradient launch (reused launch after launch)
    p->bwd_bytes = p->part_off + 2 * align_up(sizeof(float) * (size_t)kWideSlicesMax * hp * hp, 256)       // (W_l | W_r partials
                   + align_up(sizeof(float) * (size_t)kWideSlicesMax * hp, 256);                            //  | bias partials)
    return HEXGNN_OK;
}

struct WidePackArgs {
    const float* wl[kMaxLayers];
    const float* bl[kMaxLayers];
    const float* wr[kMaxLayers];
    size_t w_off[kMaxLayers], bias_off[kMaxLayers];
    int hp, hidden, c_in, small_first;
};

__global__ void wide_pack_kernel(WidePackArgs a, char* __restrict__ wpack) {
    const int l = blockIdx.y, hp = a.hp, H = a.hidden;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    float* bias = (float*)(wpack + a.bias_off[l]);
    if (tid < hp) bias[tid] = tid < H ? a.bl[l][tid] : 0.f;
    if (l == 0 && a.small_first) {
        float* w0 = (float*)(wpack + a.w_off[l]);
        const int tot = hp * kSmallCin;
        if (tid < tot) {
            const int o = tid / kSmallCin, q = tid % kSmallCin;
            const bool ok = o < H && q < a.c_in;
            w0[tid] = ok ? a.wl[l][o * a.c_in + q] : 0.f;
            w0[tot + tid] = ok ? a.wr[l][o * a.c_in + q] : 0.f;
        }
        return;
    }
    if (tid >= hp * hp) return;
    const int r = tid / hp, c = tid % hp;
    float* w = (float*)(wpack + a.w_off[l]);
    const size_t sq = (size_t)hp * hp;
    const bool ok = r < H && c < H;
    w[tid] = ok ? a.wl[l][r * H + c] : 0.f;                 // WlP[o = r][k = c]
    w[sq + tid] = ok ? a.wr[l][r * H + c] : 0.f;            // WrP
    w[2 * sq + tid] = ok ? a.wl[l][c * H + r] : 0.f;        // WlT[i = r][o = c]
    w[3 * sq + tid] = ok ? a.wr[l][c * H + r] : 0.f;        // WrT
}

// ---- raw first layer (c_in <= 8) ------------------------------------------------------------------------------------------
__global__ void wide_first_agg_kernel(int n, int c_in, const int* __restrict__ rowptr, const int* __restrict__ col,
                                      const float* __restrict__ invdeg, const float* __restrict__ x, int xs,
                                      float* __restrict__ agg0 /*[n][8]*/) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n) return;
    float a[kSmallCin];
#pragma unroll
    for (int q = 0; q < kSmallCin; ++q) a[q] = 0.f;
    for (int e = rowptr[row]; e < rowptr[row + 1]; ++e) {
        const float* xr = x + (size_t)col[e] * xs;
#pragma unroll
        for (int q = 0; q < kSmallCin; ++q) if (q < c_in) a[q] += xr[q];
    }
    const float sc = invdeg[row];
#pragma unroll
    for (int q = 0; q < kSmallCin; ++q) agg0[(size_t)row * kSmallCin + q] = a[q] * sc;
}

__global__ void wide_first_out_kernel(int n, int c_in, int hp, const float* __restrict__ agg0, const float* __restrict__ x,
                                      int xs, const float* __restrict__ w0, const float* __restrict__ bias,
                                      float* __restrict__ y, int relu) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n * hp) return;
    const int row = (int)(i / hp), c = (int)(i % hp);
    const float* wl = w0 + (size_t)c * kSmallCin;
    const float* wr = w0 + (size_t)hp * kSmallCin + (size_t)c * kSmallCin;
    float v = bias[c];
#pragma unroll
    for (int q = 0; q < kSmallCin; ++q) {
        const float xq = q < c_in ? x[(size_t)row * xs + q] : 0.f;
        v += wl[q] * agg0[(size_t)row * kSmallCin + q] + wr[q] * xq;
    }
    y[i] = (v > 0.f || !relu) ? v : 0.f;
}

// d w0: thread (o, q, which) walks the rows of one slice in order; the slices are added in order afterwards (deterministic)
__global__ __launch_bounds__(256) void wide_first_dw_kernel(int n, int c_in, int H, int hp, int rows_per_slice,
                                                           const float* __restrict__ G, const float* __restrict__ agg0,
                                                           const float* __restrict__ x, int xs,
                                                           float* __restrict__ part /*[S][HP][16]*/) {
    const int o = blockIdx.x * 16 + (threadIdx.x >> 4), qq = threadIdx.x & 15, sl = blockIdx.y;
    const int which = qq >> 3, q = qq & 7;
    const int m_lo = sl * rows_per_slice, m_hi = min(n, m_lo + rows_per_slice);
    float acc = 0.f;
    if (o < H && q < c_in) {
        for (int m = m_lo; m < m_hi; ++m) {
            const float g = G[(size_t)m * hp + o];
            const float v = which == 0 ? agg0[(size_t)m * kSmallCin + q] : x[(size_t)m * xs + q];
            acc += g * v;
        }
    }
    if (o < hp) part[((size_t)sl * hp + o) * 16 + qq] = acc;
}
__global__ void wide_first_dw_reduce_kernel(int S, int c_in, int H, int hp, const float* __restrict__ part,
                                            float* __restrict__ d_wl, float* __restrict__ d_wr) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= H * 16) return;
    const int o = idx / 16, qq = idx % 16, which = qq >> 3, q = qq & 7;
    if (q >= c_in) return;
    float s = 0.f;
    for (int k = 0; k < S; ++k) s += part[((size_t)k * hp + o) * 16 + qq];
    (which == 0 ? d_wl : d_wr)[o * c_in + q] = s;
}

// ---- mean gather: out[i] = s_i * sum_{j in N(i)} x[j]  (16-byte column groups) ----------------------------------------------
__global__ void wide_gather_kernel(int n, int hp, const int* __restrict__ rowptr, const int* __restrict__ col,
                                   const float* __restrict__ scale, const float* __restrict__ x, float* __restrict__ out) {
    const int q4 = hp / 4;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n * q4) return;
    const int row = (int)(i / q4), p = (int)(i % q4);
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int e = rowptr[row]; e < rowptr[row + 1]; ++e) v += reinterpret_cast<const f32x4*>(x + (size_t)col[e] * hp)[p];
    reinterpret_cast<f32x4*>(out + (size_t)row * hp)[p] = v * scale[row];
}

// ---- C[m][n] = epi( sum_k A1[m][k] W1[n][k] (+ sum_k A2[m][k] W2[n][k]) + bias[n] ) * rowscale[m] --------------------------
// All operands in the padded layout (row stride HP, pad columns / rows zero).  Block = 4 waves = 64 rows x 64 columns; a
// wave = 16 rows x four 16-column tiles; k in chunks of 16 (one 16-byte load per lane and operand, four MFMAs per tile).
// Operands swapped (a = weight fragment, b = row fragment): lane (i, kk) ends with C[m0 + i][n0 + 4 kk .. + 3].
__global__ __launch_bounds__(256) void wide_gemm_kernel(int M, int hp, const float* __restrict__ A1, const float* __restrict__ W1,
                                                       const float* __restrict__ A2, const float* __restrict__ W2,
                                                       const float* __restrict__ bias, const float* __restrict__ rowscale,
                                                       int relu, float* __restrict__ C) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, kk = lane >> 4;
    const int m0 = blockIdx.x * 64 + wave * 16, nb = blockIdx.y * 64;
    const int row = m0 + i;
    const bool rv = row < M;
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int pass = 0; pass < 2; ++pass) {
        const float* A = pass == 0 ? A1 : A2;
        const float* W = pass == 0 ? W1 : W2;
        if (!A) continue;
        for (int k0 = 0; k0 < hp; k0 += 16) {
            const f32x4 a4 = rv ? *reinterpret_cast<const f32x4*>(A + (size_t)row * hp + k0 + 4 * kk) : z4;
            f32x4 w4[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int nn = nb + 16 * t + i;
                w4[t] = nn < hp ? *reinterpret_cast<const f32x4*>(W + (size_t)nn * hp + k0 + 4 * kk) : z4;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = mfma16x16x4(w4[t][j], a4[j], acc[t]);
            }
        }
    }
    if (!rv) return;
    const float rs = rowscale ? rowscale[row] : 1.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int nn = nb + 16 * t + 4 * kk;
        if (nn >= hp) continue;
        f32x4 v = acc[t];
        if (bias) v += *reinterpret_cast<const f32x4*>(bias + nn);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = ((v[q] > 0.f || !relu) ? v[q] : 0.f) * rs;
        *reinterpret_cast<f32x4*>(C + (size_t)row * hp + nn) = v;
    }
}

// ---- weight gradient: row-slice partials part[s][o][i] (wide_dw_tiled_kernel below), added in slice order by
// wide_slices_reduce_kernel: deterministic.
constexpr int kWideSlices = 64;
// roles (blockIdx.y): 0 = W_l partials, 1 = W_r partials (part + part_stride), 2 = bias partials [S][HP] (part + 2 part_stride).
// Eight independent loads per step of the slice loop (the one-load-per-iteration form took 16 us for 64 slices: a latency chain).
__global__ void wide_slices_reduce_kernel(int S, int H, int hp, const float* __restrict__ part, size_t part_stride,
                                          float* __restrict__ dWl, float* __restrict__ dWr, float* __restrict__ db) {
    const int role = blockIdx.y;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const float* base;
    size_t stride;
    float* out;
    if (role == 2) {
        if (idx >= H || !db) return;
        base = part + 2 * part_stride + idx; stride = (size_t)hp; out = db + idx;
    } else {
        if (idx >= H * H) return;
        const int o = idx / H, c = idx % H;
        base = part + (size_t)role * part_stride + (size_t)o * hp + c; stride = (size_t)hp * hp; out = (role ? dWr : dWl) + idx;
    }
    float s = 0.f;
    for (int k = 0; k < S; k += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = k + j < S ? base[(size_t)(k + j) * stride] : 0.f;
        s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    *out = s;
}

// column sums of G (the bias gradient): block = 16 columns x 16 row phases over one row slice, fixed-order combine
__global__ __launch_bounds__(256) void wide_colsum_kernel(int n, int hp, int rows_per_slice, const float* __restrict__ G,
                                                         float* __restrict__ part /*[S][HP]*/) {
    __shared__ float s[16][17];
    const int c = blockIdx.x * 16 + (threadIdx.x & 15), ph = threadIdx.x >> 4, sl = blockIdx.y;
    const int m_lo = sl * rows_per_slice, m_hi = min(n, m_lo + rows_per_slice);
    float acc = 0.f;
    if (c < hp) {
        for (int m = m_lo + ph; m < m_hi; m += 16) acc += G[(size_t)m * hp + c];
    }
    s[ph][threadIdx.x & 15] = acc;
    __syncthreads();
    if (ph == 0 && c < hp) {
        float t = 0.f;
        for (int p = 0; p < 16; ++p) t += s[p][threadIdx.x & 15];
        part[(size_t)sl * hp + c] = t;
    }
}
__global__ void wide_colsum_reduce_kernel(int S, int H, int hp, const float* __restrict__ part, float* __restrict__ d_b) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= H) return;
    float s = 0.f;
    for (int k = 0; k < S; ++k) s += part[(size_t)k * hp + c];
    d_b[c] = s;
}

// out = a * [y > 0] (or a)
__global__ void wide_mask_kernel(int64_t tot4, const float* __restrict__ a, const float* __restrict__ ymask,
                                 float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= tot4) return;
    f32x4 v = reinterpret_cast<const f32x4*>(a)[i];
    if (ymask) {
        const f32x4 y = reinterpret_cast<const f32x4*>(ymask)[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = y[j] > 0.f ? v[j] : 0.f;
    }
    reinterpret_cast<f32x4*>(out)[i] = v;
}

// out[i] = (dxs[i] + sum_{j in T(i)} dagg[j]) * [y_i > 0]   (dagg rows already carry their 1 / deg_j)
__global__ void wide_combine_kernel(int n, int hp, const int* __restrict__ rowptr_t, const int* __restrict__ col_t,
                                    const float* __restrict__ dxs, const float* __restrict__ dagg,
                                    const float* __restrict__ ymask, float* __restrict__ out) {
    const int q4 = hp / 4;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n * q4) return;
    const int row = (int)(i / q4), p = (int)(i % q4);
    f32x4 v = reinterpret_cast<const f32x4*>(dxs + (size_t)row * hp)[p];
    for (int e = rowptr_t[row]; e < rowptr_t[row + 1]; ++e)
        v += reinterpret_cast<const f32x4*>(dagg + (size_t)col_t[e] * hp)[p];
    if (ymask) {
        const f32x4 yv = reinterpret_cast<const f32x4*>(ymask + (size_t)row * hp)[p];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = yv[j] > 0.f ? v[j] : 0.f;
    }
    reinterpret_cast<f32x4*>(out + (size_t)row * hp)[p] = v;
}

// ---- the same product, LDS-tiled (round 4) ---------------------------------------------------------------------------------
// Block = 8 waves x 16 rows = 128 rows x ALL HP = 16 NTW columns.  K runs in chunks of 16 over [A1 | A2]: the chunk of the weight
// operand (HP columns x 16 k, 16 KB at HP = 256) is staged in LDS once for the eight waves (double buffered, one barrier per
// chunk; row stride 20 floats: the b128 fragment reads of 8 consecutive lanes cover the 32 banks), a wave's own row fragment
// (16 B per lane and chunk) is prefetched in registers one chunk ahead.  Per chunk and wave: NTW b128 LDS reads, 4 NTW MFMAs
// (tile-interleaved: consecutive MFMAs never share an accumulator).  Same operand swap and k order as wide_gemm_kernel.
template <int NTW>
__global__ __launch_bounds__(512) void wide_gemm_tiled_kernel(int M, const float* __restrict__ A1, const float* __restrict__ W1,
                                                             const float* __restrict__ A2, const float* __restrict__ W2,
                                                             const float* __restrict__ bias, const float* __restrict__ rowscale,
                                                             int relu, float* __restrict__ C) {
    constexpr int HP = 16 * NTW, WS = 20;
    constexpr int kPieces = HP * 4;                       // b128 pieces of one weight chunk
    constexpr int kPer = (kPieces + 511) / 512;
    __shared__ __attribute__((aligned(16))) float Ws[2][HP * WS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, kk = lane >> 4;
    const int row = blockIdx.x * 128 + wave * 16 + i;
    const bool rv = row < M;
    const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 acc[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[t] = z4;
    const int nchunks = (A2 ? 2 : 1) * NTW;
    f32x4 wreg[kPer];
    auto load_w = [&](int c) {
        const float* W = c < NTW ? W1 : W2;
        const int k0 = (c < NTW ? c : c - NTW) * 16;
#pragma unroll
        for (int r = 0; r < kPer; ++r) {
            const int p = tid + 512 * r;
            if (p < kPieces) wreg[r] = *reinterpret_cast<const f32x4*>(W + (size_t)(p >> 2) * HP + k0 + 4 * (p & 3));
        }
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int r = 0; r < kPer; ++r) {
            const int p = tid + 512 * r;
            if (p < kPieces) *reinterpret_cast<f32x4*>(&Ws[buf][(p >> 2) * WS + 4 * (p & 3)]) = wreg[r];
        }
    };
    auto load_a = [&](int c) -> f32x4 {
        const float* A = c < NTW ? A1 : A2;
        const int k0 = (c < NTW ? c : c - NTW) * 16;
        return rv ? *reinterpret_cast<const f32x4*>(A + (size_t)row * HP + k0 + 4 * kk) : z4;
    };
    load_w(0);
    f32x4 a_cur = load_a(0);
    store_w(0);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        f32x4 a_nxt = z4;
        if (c + 1 < nchunks) { load_w(c + 1); a_nxt = load_a(c + 1); }
        f32x4 w4[NTW];
#pragma unroll
        for (int t = 0; t < NTW; ++t) w4[t] = *reinterpret_cast<const f32x4*>(&Ws[buf][(16 * t + i) * WS + 4 * kk]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int t = 0; t < NTW; ++t) acc[t] = mfma16x16x4(w4[t][j], a_cur[j], acc[t]);
        }
        if (c + 1 < nchunks) store_w(buf ^ 1);            // (its readers passed the previous barrier)
        a_cur = a_nxt;
        __syncthreads();
    }
    if (!rv) return;
    const float rs = rowscale ? rowscale[row] : 1.f;
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        const int nn = 16 * t + 4 * kk;
        f32x4 v = acc[t];
        if (bias) v += *reinterpret_cast<const f32x4*>(bias + nn);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = ((v[q] > 0.f || !relu) ? v[q] : 0.f) * rs;
        *reinterpret_cast<f32x4*>(C + (size_t)row * HP + nn) = v;
    }
}

static void gemm(int n, int hp, const float* A1, const float* W1, const float* A2, const float* W2, const float* bias,
                 const float* rowscale, int relu, float* C, hipStream_t st) {
    const unsigned grid = (unsigned)((n + 127) / 128);
    switch (hp / 16) {
#define HEXGNN_WIDE_GEMM(NTW_) case NTW_: wide_gemm_tiled_kernel<NTW_><<<grid, 512, 0, st>>>(n, A1, W1, A2, W2, bias, rowscale, relu, C); return;
        HEXGNN_WIDE_GEMM(9) HEXGNN_WIDE_GEMM(10) HEXGNN_WIDE_GEMM(11) HEXGNN_WIDE_GEMM(12)
        HEXGNN_WIDE_GEMM(13) HEXGNN_WIDE_GEMM(14) HEXGNN_WIDE_GEMM(15) HEXGNN_WIDE_GEMM(16)
#undef HEXGNN_WIDE_GEMM
        default: break;
    }
    wide_gemm_kernel<<<dim3((n + 63) / 64, (hp + 63) / 64), 256, 0, st>>>(n, hp, A1, W1, A2, W2, bias, rowscale, relu, C);
}

// ---- weight gradient, LDS-tiled (round 4): part[which][s][o][i] = sum_{m in slice s} G[m][o] X_which[m][i] ---------------------
// grid (2 operands x 2 halves of the i tiles, S row slices): blockIdx.x picks the operand (agg -> W_l partials, layer input ->
// W_r partials) and the half.  Block = NT = HP / 16 waves: wave w owns output channels 16 w .. 16 w + 15 (ALL of them: G is staged
// once per block) against the half's NTI = ceil(NT / 2) i tiles.  Rows in chunks of 16, both operands through LDS (double
// buffered, one barrier per chunk), ascending rows: deterministic.  (First tiled version, 128 x 128 output blocks: 85 + 48 us per
// layer at hidden 160 -- the partial blocks idled six of eight waves; one wave per 16 x 16 tile before that.)
template <int NT>
__global__ __launch_bounds__((64 * NT)) void wide_dw_tiled_kernel(int n, int rows_per_slice, const float* __restrict__ G,
                                                                 const float* __restrict__ Xagg, const float* __restrict__ Xin,
                                                                 float* __restrict__ part, size_t part_stride /* floats per operand */) {
    // (+ the bias gradient's partials, the column sums of G over the slice, by the blocks of operand 0 / half 0: [S][HP] behind the
    // two operands' regions)
    constexpr int HP = 16 * NT, NTI = (NT + 1) / 2;
    constexpr int GSW = HP % 32 == 0 ? HP + 16 : HP + 32;               // LDS row strides (floats): == 16 (mod 32)
    constexpr int XW = 16 * NTI, XSW = XW % 32 == 0 ? XW + 16 : XW + 32;
    __shared__ __attribute__((aligned(16))) float Gs[2][16 * GSW];
    __shared__ __attribute__((aligned(16))) float Xs[2][16 * XSW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int which = blockIdx.x >> 1, half = blockIdx.x & 1, sl = blockIdx.y;
    const int i0 = half * XW;
    const float* __restrict__ X = which == 0 ? Xagg : Xin;
    const int m_lo = sl * rows_per_slice, m_hi = min(n, m_lo + rows_per_slice);
    f32x4 acc[NTI];
#pragma unroll
    for (int t = 0; t < NTI; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool do_b = blockIdx.x == 0;
    float bsum = 0.f;
    // staging: G -- every thread 16 B (16 rows x HP / 4 pieces = 64 NT); X -- the first 64 NTI threads
    constexpr int GQ = HP / 4, XQ = XW / 4;
    const int gr = tid / GQ, gq = tid % GQ;
    const bool xth = tid < 16 * XQ;
    const int xr = tid / XQ, xq = tid % XQ;
    const bool xin_range = xth && i0 + 4 * xq < HP;
    const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 rg = z4, rx = z4;
    auto issue = [&](int mc) {
        rg = mc + gr < m_hi ? *reinterpret_cast<const f32x4*>(G + (size_t)(mc + gr) * HP + 4 * gq) : z4;
        rx = (xin_range && mc + xr < m_hi) ? *reinterpret_cast<const f32x4*>(X + (size_t)(mc + xr) * HP + i0 + 4 * xq) : z4;
    };
    auto stage = [&](int buf) {
        *reinterpret_cast<f32x4*>(&Gs[buf][gr * GSW + 4 * gq]) = rg;
        if (xth) *reinterpret_cast<f32x4*>(&Xs[buf][xr * XSW + 4 * xq]) = rx;
    };
    if (m_lo < m_hi) { issue(m_lo); stage(0); }
    __syncthreads();
    int buf = 0;
    for (int mc = m_lo; mc < m_hi; mc += 16, buf ^= 1) {
        if (mc + 16 < m_hi) issue(mc + 16);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const float a = Gs[buf][(4 * ks + kq) * GSW + 16 * wave + m];
            if (do_b) bsum += a;
#pragma unroll
            for (int t = 0; t < NTI; ++t) acc[t] = mfma16x16x4(a, Xs[buf][(4 * ks + kq) * XSW + 16 * t + m], acc[t]);
        }
        if (mc + 16 < m_hi) stage(buf ^ 1);
        __syncthreads();
    }
    if (do_b) {
        bsum += __shfl_xor(bsum, 16);
        bsum += __shfl_xor(bsum, 32);
        if (kq == 0) part[2 * part_stride + (size_t)sl * HP + 16 * wave + m] = bsum;
    }
    // acc[t][r] = tile[16 wave + 4 kq + r][i0 + 16 t + m]
    float* base = part + (size_t)which * part_stride + ((size_t)sl * HP + 16 * wave + 4 * kq) * HP + i0 + m;
#pragma unroll
    for (int t = 0; t < NTI; ++t) {
        if (i0 + 16 * t < HP) {
#pragma unroll
            for (int r = 0; r < 4; ++r) base[(size_t)r * HP + 16 * t] = acc[t][r];
        }
    }
}
static bool launch_wide_dw(int n, int hp, int rps, int S, const float* G, const float* Xagg, const float* Xin, float* part,
                           size_t pstride, hipStream_t st) {
    switch (hp / 16) {
#define HEXGNN_WIDE_DW(NT_) case NT_: wide_dw_tiled_kernel<NT_><<<dim3(4, S), 64 * NT_, 0, st>>>(n, rps, G, Xagg, Xin, part, pstride); return true;
        HEXGNN_WIDE_DW(9) HEXGNN_WIDE_DW(10) HEXGNN_WIDE_DW(11) HEXGNN_WIDE_DW(12)
        HEXGNN_WIDE_DW(13) HEXGNN_WIDE_DW(14) HEXGNN_WIDE_DW(15) HEXGNN_WIDE_DW(16)
#undef HEXGNN_WIDE_DW
        default: return false;
    }
}


// ---- SAGE stack, forward ------------------------------------------------------------------------------------------------
int wide_stack_forward(int n, int c_in, int hidden, int L, const int* rowptr, const int* col, const float* invdeg,
                       const float* x, int x_stride, const float* const* wl, const float* const* bl,
                       const float* const* wr, void* wpack, float* acts, void* saved, int flags, hipStream_t st) {
    WidePlan p;
    int rc = wide_make_plan(n, c_in, hidden, L, &p);
    if (rc != HEXGNN_OK) return rc;
    if (!saved && n > 0) return HEXGNN_EINVAL;           // (the aggregate of a layer is materialised: always needs `saved`)
    if (p.small_first ? x_stride < c_in : x_stride != p.hp) return HEXGNN_EINVAL;
    WidePackArgs pa;
    for (int l = 0; l < L; ++l) {
        if (!wl[l] || !bl[l] || !wr[l]) return HEXGNN_EINVAL;
        pa.wl[l] = wl[l]; pa.bl[l] = bl[l]; pa.wr[l] = wr[l];
        pa.w_off[l] = p.w_off[l]; pa.bias_off[l] = p.bias_off[l];
    }
    pa.hp = p.hp; pa.hidden = hidden; pa.c_in = c_in; pa.small_first = p.small_first;
    wide_pack_kernel<<<dim3((p.hp * p.hp + 255) / 256, L), 256, 0, st>>>(pa, (char*)wpack);
    if (n == 0) return check_launch();
    const int hp = p.hp;
    const size_t slab = (size_t)n * hp, sq = (size_t)hp * hp;
    char* wp = (char*)wpack;
    char* sv = (char*)saved;
    const unsigned g4 = (unsigned)(((int64_t)n * (hp / 4) + 255) / 256);
    for (int l = 0; l < L; ++l) {
        float* y = acts + slab * l;
        const float* bias = (const float*)(wp + p.bias_off[l]);
        const int relu = !(l == L - 1 && (flags & HEXGNN_SAGE_LINEAR_LAST));
        if (l == 0 && p.small_first) {
            float* agg0 = (float*)(sv + p.agg_off[0]);
            wide_first_agg_kernel<<<(n + 255) / 256, 256, 0, st>>>(n, c_in, rowptr, col, invdeg, x, x_stride, agg0);
            wide_first_out_kernel<<<(unsigned)(((int64_t)n * hp + 255) / 256), 256, 0, st>>>(
                n, c_in, hp, agg0, x, x_stride, (const float*)(wp + p.w_off[0]), bias, y, relu);
        } else {
            const float* xin = l == 0 ? x : acts + slab * (l - 1);
            float* agg = (float*)(sv + p.agg_off[l]);
            const float* w = (const float*)(wp + p.w_off[l]);
            wide_gather_kernel<<<g4, 256, 0, st>>>(n, hp, rowptr, col, invdeg, xin, agg);
            gemm(n, hp, agg, w, xin, w + sq, bias, nullptr, relu, y, st);
        }
    }
    return check_launch();
}

// ---- SAGE stack, backward -------------------------------------------------------------------------------------------------
int wide_stack_backward(int n, int c_in, int hidden, int L, const int* rowptr_t, const int* col_t, const float* invdeg,
                        const float* x, int x_stride, const float* acts, const void* saved, const void* wpack,
                        const float* dy, float* dx, float* const* d_wl, float* const* d_bl, float* const* d_wr,
                        void* workspace, size_t workspace_bytes, int flags, int tap_layer, float* tap_out, hipStream_t st) {
    WidePlan p;
    int rc = wide_make_plan(n, c_in, hidden, L, &p);
    if (rc != HEXGNN_OK) return rc;
    if (!workspace || workspace_bytes < p.bwd_bytes) return HEXGNN_EWORKSPACE;
    for (int l = 0; l < L; ++l) if (!d_wl[l] || !d_bl[l] || !d_wr[l]) return HEXGNN_EINVAL;
    if (n == 0) {
        for (int l = 0; l < L; ++l) {
            const int in = l == 0 ? c_in : hidden;
            (void)hipMemsetAsync(d_wl[l], 0, sizeof(float) * (size_t)hidden * in, st);
            (void)hipMemsetAsync(d_wr[l], 0, sizeof(float) * (size_t)hidden * in, st);
            (void)hipMemsetAsync(d_bl[l], 0, sizeof(float) * (size_t)hidden, st);
        }
        return check_launch();
    }
    const int hp = p.hp;
    const size_t slab = (size_t)n * hp, sq = (size_t)hp * hp;
    const size_t slab_b = align_up(sizeof(float) * slab, 256);
    char* ws = (char*)workspace;
    const char* wp = (const char*)wpack;
    const char* sv = (const char*)saved;
    auto Gl = [&](int l) { return (float*)(ws + p.g_off) + slab * l; };     // (contiguous [L][n][HP], as the narrow plan)
    float* dagg = (float*)(ws + p.tmp_off);
    float* dxs = (float*)(ws + p.tmp_off + slab_b);
    const int64_t tot4 = (int64_t)n * (hp / 4);
    const unsigned g4 = (unsigned)((tot4 + 255) / 256);
    if (flags & HEXGNN_SAGE_DY_IN_PLACE) {
        if (dy != Gl(L - 1)) return HEXGNN_EINVAL;
    } else {
        const bool relu_top = !(flags & HEXGNN_SAGE_LINEAR_LAST);
        wide_mask_kernel<<<g4, 256, 0, st>>>(tot4, dy, relu_top ? acts + slab * (L - 1) : nullptr, Gl(L - 1));
    }
    if (tap_out && (tap_layer < 0 || tap_layer >= L - 1)) return HEXGNN_EINVAL;
    const int first_hidden = p.small_first ? 1 : 0;
    for (int l = L - 1; l >= first_hidden; --l) {
        float* out = l >= 1 ? Gl(l - 1) : dx;
        if (!out) break;
        const float* w = (const float*)(wp + p.w_off[l]);
        // d agg_j = (G_j Wl) / deg_j,  d xs = G Wr   (operands: the transposed copies, rows = input features)
        gemm(n, hp, Gl(l), w + 2 * sq, nullptr, nullptr, nullptr, invdeg, 0, dagg, st);
        gemm(n, hp, Gl(l), w + 3 * sq, nullptr, nullptr, nullptr, nullptr, 0, dxs, st);
        if (tap_out && l - 1 == tap_layer) wide_combine_kernel<<<g4, 256, 0, st>>>(n, hp, rowptr_t, col_t, dxs, dagg, nullptr, tap_out);
        wide_combine_kernel<<<g4, 256, 0, st>>>(n, hp, rowptr_t, col_t, dxs, dagg, l >= 1 ? acts + slab * (l - 1) : nullptr, out);
    }
    // parameter gradients: row-slice partials (one scratch region, launches are stream-ordered), added in slice order
    const int S = kWideSlices, rps = ((n + S - 1) / S + 15) / 16 * 16;
    float* part = (float*)(ws + p.part_off);
    const unsigned rg = (unsigned)((hidden * hidden + 255) / 256);
    const size_t pstride = align_up(sizeof(float) * (size_t)kWideSlicesMax * hp * hp, 256) / sizeof(float);
    for (int l = first_hidden; l < L; ++l) {
        const float* xin = l == 0 ? x : acts + slab * (l - 1);
        if (!launch_wide_dw(n, hp, rps, S, Gl(l), (const float*)(sv + p.agg_off[l]), xin, part, pstride, st)) return HEXGNN_EUNSUPPORTED;
        wide_slices_reduce_kernel<<<dim3(rg, 3), 256, 0, st>>>(S, hidden, hp, part, pstride, d_wl[l], d_wr[l], d_bl[l]);
    }
    if (p.small_first) {
        // (its threads walk a slice's rows one by one: many short slices -- 400 us with 32 slices of 984 rows)
        const int S1 = 256, rps1 = (n + S1 - 1) / S1;
        wide_first_dw_kernel<<<dim3((hp + 15) / 16, S1), 256, 0, st>>>(n, c_in, hidden, hp, rps1, Gl(0),
                                                                      (const float*)(sv + p.agg_off[0]), x, x_stride, part);
        wide_first_dw_reduce_kernel<<<(hidden * 16 + 255) / 256, 256, 0, st>>>(S1, c_in, hidden, hp, part, d_wl[0], d_wr[0]);
        wide_colsum_kernel<<<dim3((hp + 15) / 16, S), 256, 0, st>>>(n, hp, rps, Gl(0), part);
        wide_colsum_reduce_kernel<<<(hidden + 255) / 256, 256, 0, st>>>(S, hidden, hp, part, d_bl[0]);
    }
    return check_launch();
}

// ---- head tail (advantage linear, [sum|max|min|mean] pooling, value MLP, dueling combine), one workgroup per graph -----------
// Same modes, saved layout (HeadSaved) and tie rules as head_fwd_kernel / head_bwd_kernel; generic in the width.
__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ float bsum256(float v, float* s4) {
    v = wsum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = v;
    __syncthreads();
    return s4[0] + s4[1] + s4[2] + s4[3];
}

__global__ __launch_bounds__(256) void wide_head_fwd_kernel(
    int H, int hp, int mode, const int* __restrict__ gptr, const float* __restrict__ h, const float* __restrict__ lin_w,
    const float* __restrict__ lin_b, const float* __restrict__ v0_w, const float* __restrict__ v0_b,
    const float* __restrict__ v1_w, const float* __restrict__ v1_b, float* __restrict__ q, float* __restrict__ out_v,
    float* __restrict__ adv_raw, float* __restrict__ pooled, int* __restrict__ amax, int* __restrict__ amin,
    float* __restrict__ z, float* __restrict__ vraw) {
    __shared__ float s_w[kWideMaxHidden];
    __shared__ float s_pool[4 * kWideMaxHidden];
    __shared__ float s_z[kWideMaxHidden / 2];
    __shared__ float s_red[4];
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = gptr[g], r1 = gptr[g + 1], cnt = r1 - r0;
    const int H2 = H / 2, H4 = 4 * H;
    for (int c = tid; c < H; c += 256) s_w[c] = lin_w[c];
    __syncthreads();
    const float lb = lin_b[0];
    float tsum = 0.f;
    for (int row = r0 + wave; row < r1; row += 4) {           // one wave per row
        float a = 0.f;
        for (int c = lane; c < H; c += 64) a += h[(size_t)row * hp + c] * s_w[c];
        a = wsum(a);
        if (lane == 0) {
            a += lb;
            adv_raw[row] = a;
            const float t = 2.f * tanhf(a);
            tsum += t;
            if (mode == 2) q[row] = t;
            if (mode >= 3) q[row] = a;
        }
    }
    if (mode == 2 || mode == 4) return;
    const float adv_total = bsum256(tsum, s_red);
    for (int c = tid; c < H; c += 256) {                      // pooling: a thread per column, rows in ascending order
        float sum = 0.f, mx = -INFINITY, mn = INFINITY;
        int ax = -1, an = -1;
        for (int row = r0; row < r1; ++row) {
            const float v = h[(size_t)row * hp + c];
            sum += v;
            if (v > mx) { mx = v; ax = row; }
            if (v < mn) { mn = v; an = row; }
        }
        if (cnt == 0) { mx = 0.f; mn = 0.f; }
        const float mean = sum / (float)max(cnt, 1);
        s_pool[c] = sum; s_pool[H + c] = mx; s_pool[2 * H + c] = mn; s_pool[3 * H + c] = mean;
        float* pg = pooled + (size_t)g * H4;
        pg[c] = sum; pg[H + c] = mx; pg[2 * H + c] = mn; pg[3 * H + c] = mean;
        amax[(size_t)g * H + c] = ax;
        amin[(size_t)g * H + c] = an;
    }
    __syncthreads();
    for (int k = wave; k < H2; k += 4) {                      // value MLP: one wave per hidden unit
        float p = 0.f;
        for (int c = lane; c < H4; c += 64) p += v0_w[(size_t)k * H4 + c] * s_pool[c];
        p = wsum(p);
        if (lane == 0) {
            const float zz = fmaxf(p + v0_b[k], 0.f);
            s_z[k] = zz;
            z[(size_t)g * H2 + k] = zz;
        }
    }
    __syncthreads();
    float p = 0.f;
    for (int k = tid; k < H2; k += 256) p += v1_w[k] * s_z[k];
    const float v = bsum256(p, s_red) + v1_b[0];
    if (tid == 0) vraw[g] = v;
    const float V = mode == 3 ? v : tanhf(v);
    if ((mode == 1 || mode == 3) && tid == 0) out_v[g] = V;
    if (mode == 3) return;
    const float mean_adv = adv_total / (float)max(cnt, 1);
    for (int row = r0 + tid; row < r1; row += 256) q[row] = (mode == 0 ? V : 0.f) + 2.f * tanhf(adv_raw[row]) - mean_adv;
}

__global__ __launch_bounds__(256) void wide_head_bwd_kernel(
    int H, int hp, int mode, const int* __restrict__ gptr, const float* __restrict__ h, const float* __restrict__ lin_w,
    const float* __restrict__ v0_w, const float* __restrict__ v1_w, const float* __restrict__ adv_raw,
    const int* __restrict__ amax, const int* __restrict__ amin, const float* __restrict__ z, const float* __restrict__ vraw,
    const float* __restrict__ dq, const float* __restrict__ d_out_v, float* __restrict__ dh, float* __restrict__ dadv,
    float* __restrict__ dz, float* __restrict__ dvr, float* __restrict__ lin_part, int mask_dh) {
    __shared__ float s_dp[4 * kWideMaxHidden];
    __shared__ float s_dz[kWideMaxHidden / 2];
    __shared__ float s_lw[kWideMaxHidden];
    __shared__ int s_ax[kWideMaxHidden], s_an[kWideMaxHidden];
    __shared__ float s_red[4];
    const int g = blockIdx.x, tid = threadIdx.x;
    const int r0 = gptr[g], r1 = gptr[g + 1], cnt = r1 - r0;
    const int H2 = H / 2, H4 = 4 * H;
    const bool has_value = mode != 2 && mode != 4, raw = mode >= 3;
    for (int c = tid; c < H; c += 256) s_lw[c] = lin_w[c];
    float mean_dq = 0.f;
    const float inv_cnt = 1.f / (float)max(cnt, 1);
    if (has_value) {
        float ps = 0.f;
        for (int row = r0 + tid; row < r1; row += 256) ps += dq[row];
        const float sdq = bsum256(ps, s_red);
        mean_dq = raw ? 0.f : sdq * inv_cnt;
        for (int c = tid; c < H; c += 256) { s_ax[c] = amax[(size_t)g * H + c]; s_an[c] = amin[(size_t)g * H + c]; }
        const float dV = mode == 0 ? sdq : d_out_v[g];
        const float dv = raw ? dV : dV * sech2f(vraw[g]);
        if (tid == 0) dvr[g] = dv;
        for (int k = tid; k < H2; k += 256) {
            const float d = z[(size_t)g * H2 + k] > 0.f ? v1_w[k] * dv : 0.f;
            s_dz[k] = d;
            dz[(size_t)g * H2 + k] = d;
        }
        __syncthreads();
        for (int c = tid; c < H4; c += 256) {                 // d pooled = v0_w^T dz, k ascending
            float p = 0.f;
            for (int k = 0; k < H2; ++k) p += v0_w[(size_t)k * H4 + c] * s_dz[k];
            s_dp[c] = p;
        }
    }
    __syncthreads();
    for (int row = r0 + tid; row < r1; row += 256)
        dadv[row] = raw ? dq[row] : (dq[row] - mean_dq) * 2.f * sech2f(adv_raw[row]);
    __syncthreads();
    for (int64_t i = tid; i < (int64_t)cnt * hp; i += 256) {  // dh
        const int row = r0 + (int)(i / hp), c = (int)(i % hp);
        float t = 0.f;
        if (c < H) {
            t = dadv[row] * s_lw[c];
            if (has_value) {
                t += s_dp[c] + s_dp[3 * H + c] * inv_cnt;
                if (s_ax[c] == row) t += s_dp[H + c];
                if (s_an[c] == row) t += s_dp[2 * H + c];
            }
            if (mask_dh && !(h[(size_t)row * hp + c] > 0.f)) t = 0.f;
        }
        dh[(size_t)row * hp + c] = t;
    }
    for (int c = tid; c <= hp; c += 256) {                    // per-graph partial of the advantage linear's gradient
        float acc = 0.f;
        if (c < hp) { for (int row = r0; row < r1; ++row) acc += dadv[row] * h[(size_t)row * hp + c]; }
        else { for (int row = r0; row < r1; ++row) acc += dadv[row]; }
        lin_part[(size_t)g * (hp + 1) + c] = acc;
    }
}

int wide_head_forward(int n, int b, int hidden, int mode, const int* gptr, const float* h, const float* lin_w,
                      const float* lin_b, const float* v0_w, const float* v0_b, const float* v1_w, const float* v1_b,
                      float* q, float* out_v, void* saved, hipStream_t st) {
    const int hp = padded_width_wide(hidden);
    if (hp < 0) return HEXGNN_EUNSUPPORTED;
    if (b == 0) return HEXGNN_OK;
    const HeadSaved s = head_saved_plan(n, b, hidden);
    char* sv = (char*)saved;
    wide_head_fwd_kernel<<<b, 256, 0, st>>>(hidden, hp, mode, gptr, h, lin_w, lin_b, v0_w, v0_b, v1_w, v1_b, q, out_v,
                                            (float*)(sv + s.adv_off), (float*)(sv + s.pooled_off), (int*)(sv + s.amax_off),
                                            (int*)(sv + s.amin_off), (float*)(sv + s.z_off), (float*)(sv + s.v_off));
    return check_launch();
}

int wide_head_backward(int n, int b, int hidden, int mode, const int* gptr, const float* h, const float* lin_w,
                       const float* v0_w, const float* v1_w, const void* saved, const float* dq, const float* d_out_v,
                       float* dh, float* d_lin_w, float* d_lin_b, float* d_v0_w, float* d_v0_b, float* d_v1_w,
                       float* d_v1_b, void* workspace, size_t workspace_bytes, hipStream_t st) {
    const int hp = padded_width_wide(hidden);
    if (hp < 0) return HEXGNN_EUNSUPPORTED;
    const int mask_dh = (mode & HEXGNN_HEAD_MASK_DH) ? 1 : 0;
    mode &= ~HEXGNN_HEAD_MASK_DH;
    const HeadWs w = head_ws_plan(n, b, hidden);
    if (!workspace || workspace_bytes < w.total) return HEXGNN_EWORKSPACE;
    const HeadSaved s = head_saved_plan(n, b, hidden);
    const char* sv = (const char*)saved;
    char* ws = (char*)workspace;
    float* dadv = (float*)(ws + w.dadv_off);
    float* dz = (float*)(ws + w.dz_off);
    float* dvr = (float*)(ws + w.dvr_off);
    float* part = (float*)(ws + w.part_off);
    if (b > 0)
        wide_head_bwd_kernel<<<b, 256, 0, st>>>(hidden, hp, mode, gptr, h, lin_w, v0_w, v1_w, (const float*)(sv + s.adv_off),
                                                (const int*)(sv + s.amax_off), (const int*)(sv + s.amin_off),
                                                (const float*)(sv + s.z_off), (const float*)(sv + s.v_off), dq, d_out_v, dh,
                                                dadv, dz, dvr, part, mask_dh);
    launch_head_param_grads(b, hidden, mode, dz, dvr, (const float*)(sv + s.pooled_off), (const float*)(sv + s.z_off), part,
                            d_lin_w, d_lin_b, d_v0_w, d_v0_b, d_v1_w, d_v1_b, st);
    return check_launch();
}

}  // namespace hexgnn
