// Whole-batch LayerNorm of the --norm=True configuration.
//
// Reference: get_pre_defined("modern_two_headed") builds torch_geometric.nn.norm.LayerNorm(hidden) (GN0/models.py:8,935,945;
// pyg 2.2.0, mode="graph") and CachifiedGNN.forward calls it WITHOUT a batch vector (GN0/models.py:286-287
// `x = self.norms[i](x)`), as does DuellingTwoHeaded for after_embed_norm (GN0/models.py:550-551).  In that call form pyg
// normalises over ALL nodes and channels of the batch:   x = x - x.mean();  out = x / (x.std(unbiased=False) + eps);
// out = out * weight + bias   (eps = 1e-5 is added to the standard deviation, not the variance).  CachifiedGNN applies the
// activation after the norm, so an optional ReLU is fused here.
//
// A batch-global statistic rules out the per-graph fused kernels; this path runs on the layer-major kernels (one SAGE
// layer without ReLU, then this norm).  HBM-bound: forward = one read for the statistics + one read / one write to apply.
// Reductions are block partials in fp64 combined in a fixed order: deterministic and accurate to fp32 rounding.
#include "hexgnn_internal.h"

namespace hexgnn {

constexpr int kNormBlocks = 512;        // row-range partials (fixed: the reduction shape does not depend on the device)

struct NormWs { size_t stat_off, col_off, total; };
static NormWs norm_ws_plan(int hidden) {
    NormWs w;
    const int hp = padded_width(hidden);
    size_t off = 0;
    w.stat_off = off; off += align_up(sizeof(double) * 2 * kNormBlocks, 256);
    w.col_off = off; off += align_up(sizeof(float) * 2 * (size_t)kNormBlocks * hp, 256);
    w.total = off;
    return w;
}

__device__ __forceinline__ double block_sum_f64(double v, double* s /*[4]*/) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    return (s[0] + s[1]) + (s[2] + s[3]);
}

// partial[blk] = (sum x, sum x^2) over the block's row range, logical columns only
// (n_live, when given: a device-side row count, at most n -- a capacity-sized buffer of which only the first *n_live rows are the
// batch; both kernels then behave exactly as if launched with n = *n_live, partial shapes included)
__device__ __forceinline__ int live_rows(int n, const int* n_live) {
    if (!n_live) return n;
    const int v = *n_live;
    return v < 0 ? 0 : (v < n ? v : n);
}

__global__ __launch_bounds__(256) void norm_stats_kernel(int n, int H, int hp, const float* __restrict__ x,
                                                        double* __restrict__ partial, const int* __restrict__ n_live) {
    __shared__ double s4[4];
    n = live_rows(n, n_live);
    const int rows_per = (n + kNormBlocks - 1) / kNormBlocks;
    const int r0 = blockIdx.x * rows_per, r1 = min(n, r0 + rows_per);
    const int q4 = hp / 4;
    double s = 0.0, ss = 0.0;
    for (int i = threadIdx.x; i < (r1 - r0) * q4; i += 256) {
        const int row = r0 + i / q4, q = i % q4;
        const f32x4 v = reinterpret_cast<const f32x4*>(x + (size_t)row * hp)[q];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (4 * q + j < H) { s += (double)v[j]; ss += (double)v[j] * (double)v[j]; }
        }
    }
    s = block_sum_f64(s, s4);
    ss = block_sum_f64(ss, s4);
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = s; partial[2 * blockIdx.x + 1] = ss; }
}

// every block re-derives (mean, 1/(std+eps)) from the partials in the same fixed order, then normalises its rows
__global__ __launch_bounds__(256) void norm_apply_kernel(int n, int H, int hp, const float* __restrict__ x,
                                                        const float* __restrict__ w, const float* __restrict__ b,
                                                        float eps, int relu, const double* __restrict__ partial,
                                                        float* __restrict__ y, float* __restrict__ stats,
                                                        const int* __restrict__ n_live) {
    __shared__ double s4[4];
    __shared__ float s_mu, s_r;
    n = live_rows(n, n_live);
    double s = 0.0, ss = 0.0;
    for (int i = threadIdx.x; i < kNormBlocks; i += 256) { s += partial[2 * i]; ss += partial[2 * i + 1]; }
    s = block_sum_f64(s, s4);
    ss = block_sum_f64(ss, s4);
    if (threadIdx.x == 0) {
        const double cnt = (double)n * (double)H;
        const double mu = s / cnt;
        double var = ss / cnt - mu * mu;
        if (var < 0.0) var = 0.0;
        const float sd = (float)sqrt(var);
        s_mu = (float)mu;
        s_r = 1.f / (sd + eps);
        if (blockIdx.x == 0) { stats[0] = s_mu; stats[1] = s_r; }
    }
    __syncthreads();
    const float mu = s_mu, r = s_r;
    const int q4 = hp / 4;
    const int64_t total = (int64_t)n * q4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int row = (int)(i / q4), q = (int)(i % q4);
        const f32x4 v = reinterpret_cast<const f32x4*>(x + (size_t)row * hp)[q];
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = 4 * q + j;
            float t = 0.f;                                  // pad columns stay exactly zero
            if (c < H) {
                t = (v[j] - mu) * r * w[c] + b[c];
                if (relu) t = fmaxf(t, 0.f);
            }
            o[j] = t;
        }
        reinterpret_cast<f32x4*>(y + (size_t)row * hp)[q] = o;
    }
}

// backward pass 1: g = dy * [y > 0] * w.  Block partials: (sum g, sum g (x - mu)) in fp64 and the per-column sums
// (d_bias_c = sum dy', d_weight_c = sum dy' xhat) over the block's rows.
__global__ __launch_bounds__(256) void norm_bwd_stats_kernel(int n, int H, int hp, const float* __restrict__ x,
                                                            const float* __restrict__ y, const float* __restrict__ w,
                                                            const float* __restrict__ stats, const float* __restrict__ dy,
                                                            int relu, double* __restrict__ partial,
                                                            float* __restrict__ colpart /*[blocks][2][hp]*/) {
    constexpr int kPh = 16;                                   // row phases kept in LDS
    __shared__ double s4[4];
    __shared__ float s_col[kPh][2][128];
    const int rows_per = (n + kNormBlocks - 1) / kNormBlocks;
    const int r0 = blockIdx.x * rows_per, r1 = min(n, r0 + rows_per);
    const float mu = stats[0], r = stats[1];
    // thread = (16-byte column group q, row phase ph): three 16-byte loads per row instead of three scalars per element (the
    // scalar form read 42 MB in 28 us); fixed shape, so the sums are reproducible
    const int q4 = hp / 4;
    const int nph = min(kPh, 256 / q4);
    const int q = threadIdx.x % q4, ph = threadIdx.x / q4;
    double sg = 0.0, sgx = 0.0;
    f32x4 db = f32x4{0.f, 0.f, 0.f, 0.f}, dw = db;
    if (ph < nph) {
        f32x4 wc;
#pragma unroll
        for (int j = 0; j < 4; ++j) wc[j] = 4 * q + j < H ? w[4 * q + j] : 0.f;
        for (int row = r0 + ph; row < r1; row += nph) {
            const size_t o = (size_t)row * q4 + q;
            f32x4 d = reinterpret_cast<const f32x4*>(dy)[o];
            const f32x4 xv = reinterpret_cast<const f32x4*>(x)[o];
            if (relu) {
                const f32x4 yv = reinterpret_cast<const f32x4*>(y)[o];
#pragma unroll
                for (int j = 0; j < 4; ++j) d[j] = yv[j] > 0.f ? d[j] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (4 * q + j < H) {
                    const float xc = xv[j] - mu;
                    db[j] += d[j];
                    dw[j] += d[j] * (xc * r);
                    const float g = d[j] * wc[j];
                    sg += (double)g;
                    sgx += (double)g * (double)xc;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { s_col[ph][0][4 * q + j] = db[j]; s_col[ph][1][4 * q + j] = dw[j]; }
    }
    sg = block_sum_f64(sg, s4);
    sgx = block_sum_f64(sgx, s4);
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = sg; partial[2 * blockIdx.x + 1] = sgx; }
    __syncthreads();
    if (threadIdx.x < hp) {
        const int c = threadIdx.x;
        float b0 = 0.f, w0 = 0.f;
        for (int p = 0; p < nph; ++p) { b0 += s_col[p][0][c]; w0 += s_col[p][1][c]; }
        colpart[((size_t)blockIdx.x * 2 + 0) * hp + c] = b0;
        colpart[((size_t)blockIdx.x * 2 + 1) * hp + c] = w0;
    }
}

// backward pass 2: dx = r (g - mean g) - r^2 S (x - mu) / (N sigma),  S = sum g (x - mu),  sigma = 1/r - eps
__global__ __launch_bounds__(256) void norm_bwd_apply_kernel(int n, int H, int hp, const float* __restrict__ x,
                                                            const float* __restrict__ y, const float* __restrict__ w,
                                                            const float* __restrict__ stats, const float* __restrict__ dy,
                                                            float eps, int relu, const double* __restrict__ partial,
                                                            float* __restrict__ dx) {
    __shared__ double s4[4];
    __shared__ float s_gbar, s_k;
    double sg = 0.0, sgx = 0.0;
    for (int i = threadIdx.x; i < kNormBlocks; i += 256) { sg += partial[2 * i]; sgx += partial[2 * i + 1]; }
    sg = block_sum_f64(sg, s4);
    sgx = block_sum_f64(sgx, s4);
    const float mu = stats[0], r = stats[1];
    if (threadIdx.x == 0) {
        const double cnt = (double)n * (double)H;
        const double sigma = 1.0 / (double)r - (double)eps;
        s_gbar = (float)(sg / cnt);
        // sigma == 0 (constant input): torch's std has a zero sub-gradient there; keep the term out
        s_k = sigma > 0.0 ? (float)((double)r * (double)r * sgx / (cnt * sigma)) : 0.f;
    }
    __syncthreads();
    const float gbar = s_gbar, k = s_k;
    const int q4 = hp / 4;
    const int64_t total = (int64_t)n * q4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int row = (int)(i / q4), q = (int)(i % q4);
        const size_t o = (size_t)row * hp + 4 * q;
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + o);
        const f32x4 dv = *reinterpret_cast<const f32x4*>(dy + o);
        f32x4 yv = f32x4{1.f, 1.f, 1.f, 1.f};
        if (relu) yv = *reinterpret_cast<const f32x4*>(y + o);
        f32x4 out;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = 4 * q + j;
            float t = 0.f;
            if (c < H) {
                const float d = (relu && !(yv[j] > 0.f)) ? 0.f : dv[j];
                t = r * (d * w[c] - gbar) - k * (xv[j] - mu);
            }
            out[j] = t;
        }
        *reinterpret_cast<f32x4*>(dx + o) = out;
    }
}

// d_weight / d_bias: sum of the block partials per column, fixed order
__global__ __launch_bounds__(64) void norm_bwd_cols_kernel(int H, int hp, const float* __restrict__ colpart,
                                                          float* __restrict__ d_weight, float* __restrict__ d_bias) {
    const int c = blockIdx.x, which = blockIdx.y, lane = threadIdx.x;
    float s = 0.f;
    for (int b = lane; b < kNormBlocks; b += 64) s += colpart[((size_t)b * 2 + which) * hp + c];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) { if (which == 0) d_bias[c] = s; else d_weight[c] = s; }
}


// ---- CachedGraphNorm (GN0/models.py:644-670), the GraphNorm of the two_headed family --------------------------------------
// CachifiedGNN.forward calls it WITHOUT a batch vector (GN0/models.py:282-283), i.e. the whole batch is ONE graph and the
// statistics are per CHANNEL over all n nodes:
//     mean_c = (1/n) sum_i x_ic;  o = x - mean * mean_scale;  var_c = (1/n) sum_i o_ic^2;  y = weight * o / sqrt(var + eps) + bias
// With use_cache (eval mode after a set_cache forward) mean / var are the cached constants.  Column partials in fp64 over
// fixed row ranges, combined in a fixed order: deterministic; var through sum x^2 - mean^2 ms (2 - ms), in fp64.
struct ColWs { size_t part_off, coef_off, total; };
static ColWs col_ws_plan(int hidden) {
    ColWs w;
    const int hp = padded_width(hidden);
    size_t off = 0;
    w.part_off = off; off += align_up(sizeof(double) * 2 * (size_t)kNormBlocks * hp, 256);
    w.coef_off = off; off += align_up(sizeof(float) * 3 * (size_t)hp, 256);
    w.total = off;
    return w;
}

// partial[blk][which][c]: which 0 = sum_rows a, 1 = sum_rows b over the block's row range.
//   forward  (BWD = false): a = x,  b = x^2
//   backward (BWD = true) : a = g,  b = g * o   with g = dy * [y > 0] (relu), o = x - mean_c * ms_c
template <bool BWD>
__global__ __launch_bounds__(256) void colnorm_partials_kernel(int n, int H, int hp, const float* __restrict__ x,
                                                              const float* __restrict__ y, const float* __restrict__ dy,
                                                              const float* __restrict__ ms, const float* __restrict__ stats,
                                                              int relu, double* __restrict__ partial) {
    constexpr int kPh = 16;
    __shared__ double s_col[kPh][2][128];
    const int rows_per = (n + kNormBlocks - 1) / kNormBlocks;
    const int r0 = blockIdx.x * rows_per, r1 = min(n, r0 + rows_per);
    const int q4 = hp / 4;
    const int nph = min(kPh, 256 / q4);
    const int q = threadIdx.x % q4, ph = threadIdx.x / q4;
    double sa[4] = {0.0, 0.0, 0.0, 0.0}, sb[4] = {0.0, 0.0, 0.0, 0.0};
    if (ph < nph) {
        f32x4 sh = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (BWD) {
#pragma unroll
            for (int j = 0; j < 4; ++j) sh[j] = 4 * q + j < H ? stats[4 * q + j] * ms[4 * q + j] : 0.f;
        }
        for (int row = r0 + ph; row < r1; row += nph) {
            const size_t o = (size_t)row * q4 + q;
            const f32x4 xv = reinterpret_cast<const f32x4*>(x)[o];
            if constexpr (!BWD) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { sa[j] += (double)xv[j]; sb[j] += (double)xv[j] * (double)xv[j]; }
            } else {
                f32x4 d = reinterpret_cast<const f32x4*>(dy)[o];
                if (relu) {
                    const f32x4 yv = reinterpret_cast<const f32x4*>(y)[o];
#pragma unroll
                    for (int j = 0; j < 4; ++j) d[j] = yv[j] > 0.f ? d[j] : 0.f;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) { sa[j] += (double)d[j]; sb[j] += (double)d[j] * (double)(xv[j] - sh[j]); }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { s_col[ph][0][4 * q + j] = sa[j]; s_col[ph][1][4 * q + j] = sb[j]; }
    }
    __syncthreads();
    if (threadIdx.x < hp) {
        const int c = threadIdx.x;
        double a = 0.0, b = 0.0;
        for (int p = 0; p < nph; ++p) { a += s_col[p][0][c]; b += s_col[p][1][c]; }
        partial[((size_t)blockIdx.x * 2 + 0) * hp + c] = a;
        partial[((size_t)blockIdx.x * 2 + 1) * hp + c] = b;
    }
}

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// one wave per column: stats[c] = mean, stats[hp + c] = var (pad columns: 0 / 1)
__global__ __launch_bounds__(64) void colnorm_finalize_kernel(int n, int H, int hp, const double* __restrict__ partial,
                                                             const float* __restrict__ ms, float* __restrict__ stats) {
    const int c = blockIdx.x, lane = threadIdx.x;
    double s = 0.0, ss = 0.0;
    for (int b = lane; b < kNormBlocks; b += 64) {
        s += partial[((size_t)b * 2 + 0) * hp + c];
        ss += partial[((size_t)b * 2 + 1) * hp + c];
    }
    s = wave_sum_f64(s);
    ss = wave_sum_f64(ss);
    if (lane == 0) {
        if (c < H) {
            const double mu = s / (double)n, m = (double)ms[c];
            double var = ss / (double)n - mu * mu * m * (2.0 - m);
            if (var < 0.0) var = 0.0;
            stats[c] = (float)mu;
            stats[hp + c] = (float)var;
        } else {
            stats[c] = 0.f;
            stats[hp + c] = 1.f;
        }
    }
}

__global__ __launch_bounds__(256) void colnorm_apply_kernel(int n, int H, int hp, const float* __restrict__ x,
                                                           const float* __restrict__ w, const float* __restrict__ b,
                                                           const float* __restrict__ ms, const float* __restrict__ stats,
                                                           float eps, int relu, float* __restrict__ y) {
    const int q4 = hp / 4;
    const int64_t total = (int64_t)n * q4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int q = (int)(i % q4);
        const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = 4 * q + j;
            float t = 0.f;                                  // pad columns stay exactly zero
            if (c < H) {
                const float sd = sqrtf(stats[hp + c] + eps);
                t = w[c] * (v[j] - stats[c] * ms[c]) / sd + b[c];
                if (relu) t = fmaxf(t, 0.f);
            }
            o[j] = t;
        }
        reinterpret_cast<f32x4*>(y)[i] = o;
    }
}

// backward, per column from S_g = sum g, S_go = sum g o:   r = 1 / sqrt(var + eps),  d_bias = S_g,  d_weight = r S_go
//   fresh statistics:  d_o = A (g - B o),  A = w r,  B = r^2 S_go / n;   sum_i d_o = A (S_g - B S_o),  S_o = n mean (1 - ms)
//                      dx = d_o - C,  C = ms sum(d_o) / n;   d_mean_scale = -mean sum(d_o)
//   cached statistics: mean / var are constants:  dx = d_o = A g  (B = C = 0),  d_mean_scale = -mean A S_g
__global__ __launch_bounds__(64) void colnorm_bwd_finalize_kernel(int n, int H, int hp, const double* __restrict__ partial,
                                                                 const float* __restrict__ w, const float* __restrict__ ms,
                                                                 const float* __restrict__ stats, float eps, int use_cache,
                                                                 float* __restrict__ coef /*[3][hp]*/,
                                                                 float* __restrict__ d_w, float* __restrict__ d_b,
                                                                 float* __restrict__ d_ms) {
    const int c = blockIdx.x, lane = threadIdx.x;
    double sg = 0.0, sgo = 0.0;
    for (int b = lane; b < kNormBlocks; b += 64) {
        sg += partial[((size_t)b * 2 + 0) * hp + c];
        sgo += partial[((size_t)b * 2 + 1) * hp + c];
    }
    sg = wave_sum_f64(sg);
    sgo = wave_sum_f64(sgo);
    if (lane != 0) return;
    if (c >= H) { coef[c] = 0.f; coef[hp + c] = 0.f; coef[2 * hp + c] = 0.f; return; }
    const double mu = (double)stats[c], var = (double)stats[hp + c], m = (double)ms[c];
    const double r = 1.0 / sqrt(var + (double)eps);
    const double A = (double)w[c] * r;
    d_b[c] = (float)sg;
    d_w[c] = (float)(sgo * r);
    double B = 0.0, C = 0.0, sdo = A * sg;
    if (!use_cache) {
        B = r * r * sgo / (double)n;
        sdo = A * (sg - B * ((double)n * mu * (1.0 - m)));
        C = m * sdo / (double)n;
    }
    d_ms[c] = (float)(-mu * sdo);
    coef[c] = (float)A;
    coef[hp + c] = (float)B;
    coef[2 * hp + c] = (float)C;
}

__global__ __launch_bounds__(256) void colnorm_bwd_apply_kernel(int n, int H, int hp, const float* __restrict__ x,
                                                               const float* __restrict__ y, const float* __restrict__ dy,
                                                               const float* __restrict__ ms, const float* __restrict__ stats,
                                                               const float* __restrict__ coef, int relu,
                                                               float* __restrict__ dx) {
    const int q4 = hp / 4;
    const int64_t total = (int64_t)n * q4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int q = (int)(i % q4);
        const f32x4 xv = reinterpret_cast<const f32x4*>(x)[i];
        const f32x4 dv = reinterpret_cast<const f32x4*>(dy)[i];
        f32x4 yv = f32x4{1.f, 1.f, 1.f, 1.f};
        if (relu) yv = reinterpret_cast<const f32x4*>(y)[i];
        f32x4 out;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = 4 * q + j;
            float t = 0.f;
            if (c < H) {
                const float g = (relu && !(yv[j] > 0.f)) ? 0.f : dv[j];
                const float o = xv[j] - stats[c] * ms[c];
                t = coef[c] * (g - coef[hp + c] * o) - coef[2 * hp + c];
            }
            out[j] = t;
        }
        reinterpret_cast<f32x4*>(dx)[i] = out;
    }
}

}  // namespace hexgnn

using namespace hexgnn;

extern "C" {

size_t hexgnn_graph_layernorm_workspace_bytes(int hidden) {
    if (padded_width(hidden) < 0) return 0;
    return norm_ws_plan(hidden).total;
}

int hexgnn_graph_layernorm_forward(int n, int hidden, const float* x, const float* weight, const float* bias, float eps,
                                   int relu, float* y, float* stats, void* workspace, size_t workspace_bytes,
                                   hexgnn_stream_t stream_) {
    return hexgnn_graph_layernorm_forward_live(n, nullptr, hidden, x, weight, bias, eps, relu, y, stats, workspace,
                                               workspace_bytes, stream_);
}

int hexgnn_graph_layernorm_forward_live(int n, const int* n_live, int hidden, const float* x, const float* weight,
                                        const float* bias, float eps, int relu, float* y, float* stats, void* workspace,
                                        size_t workspace_bytes, hexgnn_stream_t stream_) {
    hipStream_t st = (hipStream_t)stream_;
    const int hp = padded_width(hidden);
    if (hp < 0) return HEXGNN_EUNSUPPORTED;
    if (n < 0 || !weight || !bias || !stats || (n > 0 && (!x || !y))) return HEXGNN_EINVAL;
    const NormWs w = norm_ws_plan(hidden);
    if (!workspace || workspace_bytes < w.total) return HEXGNN_EWORKSPACE;
    if (n == 0) return HEXGNN_OK;
    double* partial = (double*)((char*)workspace + w.stat_off);
    norm_stats_kernel<<<kNormBlocks, 256, 0, st>>>(n, hidden, hp, x, partial, n_live);
    const int64_t total = (int64_t)n * (hp / 4);
    int grid = (int)((total + 255) / 256);
    if (grid > 2048) grid = 2048;
    norm_apply_kernel<<<grid, 256, 0, st>>>(n, hidden, hp, x, weight, bias, eps, relu, partial, y, stats, n_live);
    return check_launch();
}

int hexgnn_graph_layernorm_backward(int n, int hidden, const float* x, const float* y, const float* weight,
                                    const float* stats, const float* dy, float eps, int relu, float* dx, float* d_weight,
                                    float* d_bias, void* workspace, size_t workspace_bytes, hexgnn_stream_t stream_) {
    hipStream_t st = (hipStream_t)stream_;
    const int hp = padded_width(hidden);
    if (hp < 0) return HEXGNN_EUNSUPPORTED;
    if (n < 0 || !weight || !stats || !d_weight || !d_bias || (n > 0 && (!x || !dy || !dx)) || (relu && n > 0 && !y))
        return HEXGNN_EINVAL;
    const NormWs w = norm_ws_plan(hidden);
    if (!workspace || workspace_bytes < w.total) return HEXGNN_EWORKSPACE;
    if (n == 0) {
        (void)hipMemsetAsync(d_weight, 0, sizeof(float) * hidden, st);
        (void)hipMemsetAsync(d_bias, 0, sizeof(float) * hidden, st);
        return check_launch();
    }
    double* partial = (double*)((char*)workspace + w.stat_off);
    float* colpart = (float*)((char*)workspace + w.col_off);
    norm_bwd_stats_kernel<<<kNormBlocks, 256, 0, st>>>(n, hidden, hp, x, y, weight, stats, dy, relu, partial, colpart);
    const int64_t total = (int64_t)n * (hp / 4);
    int grid = (int)((total + 255) / 256);
    if (grid > 2048) grid = 2048;
    norm_bwd_apply_kernel<<<grid, 256, 0, st>>>(n, hidden, hp, x, y, weight, stats, dy, eps, relu, partial, dx);
    norm_bwd_cols_kernel<<<dim3(hidden, 2), 64, 0, st>>>(hidden, hp, colpart, d_weight, d_bias);
    return check_launch();
}

size_t hexgnn_graph_colnorm_workspace_bytes(int hidden) {
    if (padded_width(hidden) < 0) return 0;
    return col_ws_plan(hidden).total;
}

int hexgnn_graph_colnorm_forward(int n, int hidden, const float* x, const float* weight, const float* bias,
                                 const float* mean_scale, float eps, int relu, int use_cache, float* y, float* stats,
                                 void* workspace, size_t workspace_bytes, hexgnn_stream_t stream_) {
    hipStream_t st = (hipStream_t)stream_;
    const int hp = padded_width(hidden);
    if (hp < 0) return HEXGNN_EUNSUPPORTED;
    if (n < 0 || !weight || !bias || !mean_scale || !stats || (n > 0 && (!x || !y))) return HEXGNN_EINVAL;
    const ColWs w = col_ws_plan(hidden);
    if (!workspace || workspace_bytes < w.total) return HEXGNN_EWORKSPACE;
    if (n == 0) return HEXGNN_OK;
    double* partial = (double*)((char*)workspace + w.part_off);
    if (!use_cache) {
        colnorm_partials_kernel<false><<<kNormBlocks, 256, 0, st>>>(n, hidden, hp, x, nullptr, nullptr, mean_scale, nullptr, 0,
                                                                   partial);
        colnorm_finalize_kernel<<<hp, 64, 0, st>>>(n, hidden, hp, partial, mean_scale, stats);
    }
    const int64_t total = (int64_t)n * (hp / 4);
    int grid = (int)((total + 255) / 256);
    if (grid > 2048) grid = 2048;
    colnorm_apply_kernel<<<grid, 256, 0, st>>>(n, hidden, hp, x, weight, bias, mean_scale, stats, eps, relu, y);
    return check_launch();
}

int hexgnn_graph_colnorm_backward(int n, int hidden, const float* x, const float* y, const float* weight,
                                  const float* mean_scale, const float* stats, const float* dy, float eps, int relu,
                                  int use_cache, float* dx, float* d_weight, float* d_bias, float* d_mean_scale,
                                  void* workspace, size_t workspace_bytes, hexgnn_stream_t stream_) {
    hipStream_t st = (hipStream_t)stream_;
    const int hp = padded_width(hidden);
    if (hp < 0) return HEXGNN_EUNSUPPORTED;
    if (n < 0 || !weight || !mean_scale || !stats || !d_weight || !d_bias || !d_mean_scale || (n > 0 && (!x || !dy || !dx)) ||
        (relu && n > 0 && !y))
        return HEXGNN_EINVAL;
    const ColWs w = col_ws_plan(hidden);
    if (!workspace || workspace_bytes < w.total) return HEXGNN_EWORKSPACE;
    if (n == 0) {
        (void)hipMemsetAsync(d_weight, 0, sizeof(float) * hidden, st);
        (void)hipMemsetAsync(d_bias, 0, sizeof(float) * hidden, st);
        (void)hipMemsetAsync(d_mean_scale, 0, sizeof(float) * hidden, st);
        return check_launch();
    }
    double* partial = (double*)((char*)workspace + w.part_off);
    float* coef = (float*)((char*)workspace + w.coef_off);
    colnorm_partials_kernel<true><<<kNormBlocks, 256, 0, st>>>(n, hidden, hp, x, y, dy, mean_scale, stats, relu, partial);
    colnorm_bwd_finalize_kernel<<<hp, 64, 0, st>>>(n, hidden, hp, partial, weight, mean_scale, stats, eps, use_cache, coef,
                                                  d_weight, d_bias, d_mean_scale);
    const int64_t total = (int64_t)n * (hp / 4);
    int grid = (int)((total + 255) / 256);
    if (grid > 2048) grid = 2048;
    colnorm_bwd_apply_kernel<<<grid, 256, 0, st>>>(n, hidden, hp, x, y, dy, mean_scale, stats, coef, relu, dx);
    return check_launch();
}

}  // extern "C"
