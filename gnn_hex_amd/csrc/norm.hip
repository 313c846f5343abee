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
__global__ __launch_bounds__(256) void norm_stats_kernel(int n, int H, int hp, const float* __restrict__ x,
                                                        double* __restrict__ partial) {
    __shared__ double s4[4];
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
                                                        float* __restrict__ y, float* __restrict__ stats) {
    __shared__ double s4[4];
    __shared__ float s_mu, s_r;
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
    hipStream_t st = (hipStream_t)stream_;
    const int hp = padded_width(hidden);
    if (hp < 0) return HEXGNN_EUNSUPPORTED;
    if (n < 0 || !weight || !bias || !stats || (n > 0 && (!x || !y))) return HEXGNN_EINVAL;
    const NormWs w = norm_ws_plan(hidden);
    if (!workspace || workspace_bytes < w.total) return HEXGNN_EWORKSPACE;
    if (n == 0) return HEXGNN_OK;
    double* partial = (double*)((char*)workspace + w.stat_off);
    norm_stats_kernel<<<kNormBlocks, 256, 0, st>>>(n, hidden, hp, x, partial);
    const int64_t total = (int64_t)n * (hp / 4);
    int grid = (int)((total + 255) / 256);
    if (grid > 2048) grid = 2048;
    norm_apply_kernel<<<grid, 256, 0, st>>>(n, hidden, hp, x, weight, bias, eps, relu, partial, y, stats);
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

}  // extern "C"
