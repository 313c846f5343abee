// Head tail: per-node advantage Linear(H,1), 4-way graph pooling [sum|max|min|mean], value MLP
// Linear(4H,H/2)->relu->Linear(H/2,1) and the dueling combine, one workgroup per graph.
//
// Reference: HeadNetwork.forward GN0/models.py:374-384, MLP GN0/models.py:36-82,
// DuellingTwoHeaded.forward GN0/models.py:567-584, torch_scatter.scatter (sum/max/min/mean).
// HBM/L2-bound (reads h once forward, writes dh once backward); no MFMA: the dense parts are
// [1 x 4H] x [4H x H/2] per graph.
#include "hexgnn_common.h"

namespace hexgnn {

struct HeadSaved {
    size_t adv_off, pooled_off, amax_off, amin_off, z_off, v_off, total;
};
static HeadSaved head_saved_plan(int n, int b, int hidden) {
    HeadSaved s;
    size_t off = 0;
    const int h2 = hidden / 2;
    s.adv_off = off; off += align_up(sizeof(float) * (size_t)n, 256);
    s.pooled_off = off; off += align_up(sizeof(float) * (size_t)b * 4 * hidden, 256);
    s.amax_off = off; off += align_up(sizeof(int) * (size_t)b * hidden, 256);
    s.amin_off = off; off += align_up(sizeof(int) * (size_t)b * hidden, 256);
    s.z_off = off; off += align_up(sizeof(float) * (size_t)b * (h2 > 0 ? h2 : 1), 256);
    s.v_off = off; off += align_up(sizeof(float) * (size_t)b, 256);
    s.total = off;
    return s;
}

struct HeadWs {
    size_t dadv_off, dz_off, dvr_off, part_off, total;
    int S, rps;
};
static HeadWs head_ws_plan(int n, int b, int hidden) {
    HeadWs w;
    const int hp = padded_width(hidden), h2 = hidden / 2;
    size_t off = 0;
    w.dadv_off = off; off += align_up(sizeof(float) * (size_t)n, 256);
    w.dz_off = off; off += align_up(sizeof(float) * (size_t)b * (h2 > 0 ? h2 : 1), 256);
    w.dvr_off = off; off += align_up(sizeof(float) * (size_t)b, 256);
    w.S = (n + 511) / 512; if (w.S < 1) w.S = 1; if (w.S > 256) w.S = 256;
    w.rps = (n + w.S - 1) / w.S;
    w.part_off = off; off += align_up(sizeof(float) * (size_t)w.S * (hp + 1), 256);
    w.total = off;
    return w;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// block = 256 threads (4 waves), one graph per block
__global__ __launch_bounds__(256) void head_fwd_kernel(
    int H, int hp, int mode, const int* __restrict__ gptr, const float* __restrict__ h,
    const float* __restrict__ lin_w, const float* __restrict__ lin_b, const float* __restrict__ v0_w,
    const float* __restrict__ v0_b, const float* __restrict__ v1_w, const float* __restrict__ v1_b,
    float* __restrict__ q, float* __restrict__ out_v, float* __restrict__ adv_raw, float* __restrict__ pooled,
    int* __restrict__ amax, int* __restrict__ amin, float* __restrict__ z, float* __restrict__ vraw) {
    __shared__ float s_pool[4 * 128];
    __shared__ float s_z[64];
    __shared__ float s_part[4];
    __shared__ float s_v;
    const int g = blockIdx.x;
    const int r0 = gptr[g], r1 = gptr[g + 1], cnt = r1 - r0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H2 = H / 2, H4 = 4 * H;

    // 1. advantages: one wave per row (lane 0 owns the row's scalar for the rest of the kernel)
    const float w0 = lane < H ? lin_w[lane] : 0.f;
    const float w1 = lane + 64 < H ? lin_w[lane + 64] : 0.f;
    const float lb = lin_b[0];
    float tsum = 0.f;
    for (int row = r0 + wave; row < r1; row += 4) {
        const float* hr = h + (size_t)row * hp;
        float p = (lane < H ? hr[lane] * w0 : 0.f) + (lane + 64 < H ? hr[lane + 64] * w1 : 0.f);
        p = wave_sum(p);
        if (lane == 0) {
            const float a = p + lb;
            adv_raw[row] = a;
            const float t = 2.f * tanhf(a);
            tsum += t;
            if (mode == 2) q[row] = t;
        }
    }
    if (mode == 2) return;
    if (lane == 0) s_part[wave] = tsum;

    // 2. pooling: thread c owns feature column c
    if (tid < H) {
        float sum = 0.f, mx = -INFINITY, mn = INFINITY;
        int ax = -1, an = -1;
        for (int row = r0; row < r1; ++row) {
            const float v = h[(size_t)row * hp + tid];
            sum += v;
            if (v > mx) { mx = v; ax = row; }
            if (v < mn) { mn = v; an = row; }
        }
        if (cnt == 0) { mx = 0.f; mn = 0.f; }
        const float mean = sum / (float)max(cnt, 1);
        s_pool[tid] = sum; s_pool[H + tid] = mx; s_pool[2 * H + tid] = mn; s_pool[3 * H + tid] = mean;
        float* pg = pooled + (size_t)g * H4;
        pg[tid] = sum; pg[H + tid] = mx; pg[2 * H + tid] = mn; pg[3 * H + tid] = mean;
        amax[(size_t)g * H + tid] = ax;
        amin[(size_t)g * H + tid] = an;
    }
    __syncthreads();

    // 3. value MLP
    for (int k = wave; k < H2; k += 4) {
        const float* wr = v0_w + (size_t)k * H4;
        float p = 0.f;
        for (int c = lane; c < H4; c += 64) p += wr[c] * s_pool[c];
        p = wave_sum(p);
        if (lane == 0) {
            const float zz = fmaxf(p + v0_b[k], 0.f);
            s_z[k] = zz;
            z[(size_t)g * H2 + k] = zz;
        }
    }
    __syncthreads();
    if (wave == 0) {
        float p = lane < H2 ? v1_w[lane] * s_z[lane] : 0.f;
        p = wave_sum(p);
        if (lane == 0) {
            const float v = p + v1_b[0];
            vraw[g] = v;
            s_v = tanhf(v);
        }
    }
    __syncthreads();

    // 4. dueling combine
    const float mean_adv = (s_part[0] + s_part[1] + s_part[2] + s_part[3]) / (float)max(cnt, 1);
    const float V = s_v;
    if (mode == 1 && tid == 0) out_v[g] = V;
    if (lane == 0) {
        for (int row = r0 + wave; row < r1; row += 4) {
            const float t = 2.f * tanhf(adv_raw[row]);  // written by this same lane above
            q[row] = (mode == 0 ? V : 0.f) + t - mean_adv;
        }
    }
}

__global__ __launch_bounds__(256) void head_bwd_kernel(
    int H, int hp, int mode, const int* __restrict__ gptr, const float* __restrict__ lin_w,
    const float* __restrict__ v0_w, const float* __restrict__ v1_w, const float* __restrict__ adv_raw,
    const int* __restrict__ amax, const int* __restrict__ amin, const float* __restrict__ z,
    const float* __restrict__ vraw, const float* __restrict__ dq, const float* __restrict__ d_out_v,
    float* __restrict__ dh, float* __restrict__ dadv, float* __restrict__ dz, float* __restrict__ dvr) {
    __shared__ float s_dp[4 * 128];
    __shared__ float s_dz[64];
    __shared__ int s_ax[128], s_an[128];
    __shared__ float s_part[4];
    const int g = blockIdx.x;
    const int r0 = gptr[g], r1 = gptr[g + 1], cnt = r1 - r0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H2 = H / 2, H4 = 4 * H;
    const float w0 = lane < H ? lin_w[lane] : 0.f;
    const float w1 = lane + 64 < H ? lin_w[lane + 64] : 0.f;

    if (mode == 2) {
        for (int row = r0 + wave; row < r1; row += 4) {
            const float t = tanhf(adv_raw[row]);
            const float dar = dq[row] * 2.f * (1.f - t * t);
            if (lane == 0) dadv[row] = dar;
            float* dr = dh + (size_t)row * hp;
            if (lane < hp) dr[lane] = dar * w0;
            if (lane + 64 < hp) dr[lane + 64] = dar * w1;
        }
        return;
    }
    // sum of dq over the graph (fixed order: 4 strided partials)
    float ps = 0.f;
    for (int row = r0 + tid; row < r1; row += 256) ps += dq[row];
    ps = wave_sum(ps);
    if (lane == 0) s_part[wave] = ps;
    if (tid < H) { s_ax[tid] = amax[(size_t)g * H + tid]; s_an[tid] = amin[(size_t)g * H + tid]; }
    __syncthreads();
    const float sdq = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    const float inv_cnt = 1.f / (float)max(cnt, 1);
    const float mean_dq = sdq * inv_cnt;
    const float dV = mode == 0 ? sdq : d_out_v[g];
    const float V = tanhf(vraw[g]);
    const float dv = dV * (1.f - V * V);
    if (tid == 0) dvr[g] = dv;
    if (tid < H2) {
        const float zz = z[(size_t)g * H2 + tid];
        const float d = zz > 0.f ? v1_w[tid] * dv : 0.f;
        s_dz[tid] = d;
        dz[(size_t)g * H2 + tid] = d;
    }
    __syncthreads();
    for (int c = tid; c < H4; c += 256) {
        float p = 0.f;
        for (int k = 0; k < H2; ++k) p += v0_w[(size_t)k * H4 + c] * s_dz[k];
        s_dp[c] = p;
    }
    __syncthreads();
    for (int row = r0 + wave; row < r1; row += 4) {
        const float t = tanhf(adv_raw[row]);
        const float dar = (dq[row] - mean_dq) * 2.f * (1.f - t * t);
        if (lane == 0) dadv[row] = dar;
        float* dr = dh + (size_t)row * hp;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int c = lane + 64 * half;
            if (c < hp) {
                float v = 0.f;
                if (c < H) {
                    v = dar * (half ? w1 : w0) + s_dp[c] + s_dp[3 * H + c] * inv_cnt;
                    if (s_ax[c] == row) v += s_dp[H + c];
                    if (s_an[c] == row) v += s_dp[2 * H + c];
                }
                dr[c] = v;
            }
        }
    }
}

// value-head parameter gradients: fixed order over graphs (deterministic)
__global__ void head_value_wgrad_kernel(int b, int H, const float* __restrict__ dz, const float* __restrict__ dvr,
                                        const float* __restrict__ pooled, const float* __restrict__ z,
                                        float* __restrict__ d_v0_w, float* __restrict__ d_v0_b,
                                        float* __restrict__ d_v1_w, float* __restrict__ d_v1_b) {
    const int H2 = H / 2, H4 = 4 * H;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int n0 = H2 * H4;
    if (idx < n0) {
        const int k = idx / H4, c = idx % H4;
        float s = 0.f;
        for (int g = 0; g < b; ++g) s += dz[(size_t)g * H2 + k] * pooled[(size_t)g * H4 + c];
        d_v0_w[idx] = s;
    } else if (idx < n0 + H2) {
        const int k = idx - n0;
        float s = 0.f;
        for (int g = 0; g < b; ++g) s += dz[(size_t)g * H2 + k];
        d_v0_b[k] = s;
    } else if (idx < n0 + 2 * H2) {
        const int k = idx - n0 - H2;
        float s = 0.f;
        for (int g = 0; g < b; ++g) s += dvr[g] * z[(size_t)g * H2 + k];
        d_v1_w[k] = s;
    } else if (idx == n0 + 2 * H2) {
        float s = 0.f;
        for (int g = 0; g < b; ++g) s += dvr[g];
        d_v1_b[0] = s;
    }
}

// advantage Linear gradient, stage 1: partial[s][c] = sum_{rows in slice} dadv[row]*h[row][c]; partial[s][hp] = sum dadv
__global__ __launch_bounds__(256) void head_lin_grad_kernel(int n, int hp, int rps, const float* __restrict__ dadv,
                                                          const float* __restrict__ h, float* __restrict__ part) {
    __shared__ float red[129];
    const int tid = threadIdx.x, c = tid & 127, ph = tid >> 7;
    const int r_beg = blockIdx.x * rps, r_end = min(n, r_beg + rps);
    float acc = 0.f, accb = 0.f;
    for (int row = r_beg + ph; row < r_end; row += 2) {
        const float d = dadv[row];
        if (c < hp) acc += d * h[(size_t)row * hp + c];
        accb += d;
    }
    if (ph == 1) { red[c] = acc; if (c == 0) red[128] = accb; }
    __syncthreads();
    if (ph == 0) {
        if (c < hp) part[(size_t)blockIdx.x * (hp + 1) + c] = acc + red[c];
        if (c == 0) part[(size_t)blockIdx.x * (hp + 1) + hp] = accb + red[128];
    }
}

__global__ void head_lin_grad_reduce_kernel(int S, int hp, int H, const float* __restrict__ part,
                                            float* __restrict__ d_lin_w, float* __restrict__ d_lin_b) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > H) return;
    const int src = c < H ? c : hp;
    float s = 0.f;
    for (int i = 0; i < S; ++i) s += part[(size_t)i * (hp + 1) + src];
    if (c < H) d_lin_w[c] = s; else d_lin_b[0] = s;
}

}  // namespace hexgnn

using namespace hexgnn;

extern "C" {

size_t hexgnn_head_saved_bytes(int n, int b, int hidden) {
    if (n < 0 || b < 0 || padded_width(hidden) < 0) return 0;
    return head_saved_plan(n, b, hidden).total;
}

int hexgnn_head_forward(int n, int b, int hidden, int mode, const int* gptr, const float* h, const float* lin_w,
                        const float* lin_b, const float* v0_w, const float* v0_b, const float* v1_w,
                        const float* v1_b, float* q, float* out_v, void* saved, hexgnn_stream_t stream_) {
    const int hp = padded_width(hidden);
    if (hp < 0 || hidden < 2) return HEXGNN_EUNSUPPORTED;
    if (n < 0 || b < 0 || mode < 0 || mode > 2) return HEXGNN_EINVAL;
    if (!gptr || !lin_w || !lin_b || !saved) return HEXGNN_EINVAL;
    if (mode != 2 && (!v0_w || !v0_b || !v1_w || !v1_b)) return HEXGNN_EINVAL;
    if (mode == 1 && !out_v) return HEXGNN_EINVAL;
    if (n > 0 && (!h || !q)) return HEXGNN_EINVAL;
    if (b == 0) return HEXGNN_OK;
    const HeadSaved s = head_saved_plan(n, b, hidden);
    char* sv = (char*)saved;
    KernelTimer kt(HEXGNN_K_HEAD_FWD, (hipStream_t)stream_);
    head_fwd_kernel<<<b, 256, 0, (hipStream_t)stream_>>>(
        hidden, hp, mode, gptr, h, lin_w, lin_b, v0_w, v0_b, v1_w, v1_b, q, out_v, (float*)(sv + s.adv_off),
        (float*)(sv + s.pooled_off), (int*)(sv + s.amax_off), (int*)(sv + s.amin_off), (float*)(sv + s.z_off),
        (float*)(sv + s.v_off));
    return check_launch();
}

size_t hexgnn_head_backward_workspace_bytes(int n, int b, int hidden) {
    if (n < 0 || b < 0 || padded_width(hidden) < 0) return 0;
    return head_ws_plan(n, b, hidden).total;
}

int hexgnn_head_backward(int n, int b, int hidden, int mode, const int* gptr, const float* h, const float* lin_w,
                         const float* v0_w, const float* v1_w, const void* saved, const float* dq,
                         const float* d_out_v, float* dh, float* d_lin_w, float* d_lin_b, float* d_v0_w,
                         float* d_v0_b, float* d_v1_w, float* d_v1_b, void* workspace, size_t workspace_bytes,
                         hexgnn_stream_t stream_) {
    hipStream_t st = (hipStream_t)stream_;
    const int hp = padded_width(hidden);
    if (hp < 0 || hidden < 2) return HEXGNN_EUNSUPPORTED;
    if (n < 0 || b < 0 || mode < 0 || mode > 2) return HEXGNN_EINVAL;
    if (!gptr || !lin_w || !saved || !d_lin_w || !d_lin_b) return HEXGNN_EINVAL;
    if (mode != 2 && (!v0_w || !v1_w || !d_v0_w || !d_v0_b || !d_v1_w || !d_v1_b)) return HEXGNN_EINVAL;
    if (mode == 1 && !d_out_v) return HEXGNN_EINVAL;
    if (n > 0 && (!h || !dq || !dh)) return HEXGNN_EINVAL;
    const HeadWs w = head_ws_plan(n, b, hidden);
    if (!workspace || workspace_bytes < w.total) return HEXGNN_EWORKSPACE;
    const HeadSaved s = head_saved_plan(n, b, hidden);
    const char* sv = (const char*)saved;
    char* ws = (char*)workspace;
    float* dadv = (float*)(ws + w.dadv_off);
    float* dz = (float*)(ws + w.dz_off);
    float* dvr = (float*)(ws + w.dvr_off);
    float* part = (float*)(ws + w.part_off);
    const int H2 = hidden / 2, H4 = 4 * hidden;
    KernelTimer kt(HEXGNN_K_HEAD_BWD, st);
    if (b > 0)
        head_bwd_kernel<<<b, 256, 0, st>>>(hidden, hp, mode, gptr, lin_w, v0_w, v1_w, (const float*)(sv + s.adv_off),
                                           (const int*)(sv + s.amax_off), (const int*)(sv + s.amin_off),
                                           (const float*)(sv + s.z_off), (const float*)(sv + s.v_off), dq, d_out_v,
                                           dh, dadv, dz, dvr);
    if (mode != 2) {
        const int tot = H2 * H4 + 2 * H2 + 1;
        head_value_wgrad_kernel<<<(tot + 255) / 256, 256, 0, st>>>(
            b, hidden, dz, dvr, (const float*)(sv + s.pooled_off), (const float*)(sv + s.z_off), d_v0_w, d_v0_b,
            d_v1_w, d_v1_b);
    }
    head_lin_grad_kernel<<<w.S, 256, 0, st>>>(n, hp, w.rps, dadv, h, part);
    head_lin_grad_reduce_kernel<<<(hidden + 1 + 127) / 128, 128, 0, st>>>(w.S, hp, hidden, part, d_lin_w, d_lin_b);
    return check_launch();
}

}  // extern "C"
