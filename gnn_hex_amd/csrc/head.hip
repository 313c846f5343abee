// Head tail: per-node advantage Linear(H,1), 4-way graph pooling [sum|max|min|mean], value MLP
// Linear(4H,H/2)->relu->Linear(H/2,1) and the dueling combine, one workgroup per graph.
//
// Reference: HeadNetwork.forward GN0/models.py:374-384, MLP GN0/models.py:36-82,
// DuellingTwoHeaded.forward GN0/models.py:567-584, torch_scatter.scatter (sum/max/min/mean).
// HBM/L2-bound (reads h once forward, writes dh once backward); no MFMA: the dense parts are
// [1 x 4H] x [4H x H/2] per graph.
#include "hexgnn_internal.h"

namespace hexgnn {

HeadSaved head_saved_plan(int n, int b, int hidden) {
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

HeadWs head_ws_plan(int n, int b, int hidden) {
    HeadWs w;
    const int hp = padded_width_wide(hidden), h2 = hidden / 2;
    size_t off = 0;
    w.dadv_off = off; off += align_up(sizeof(float) * (size_t)n, 256);
    w.dz_off = off; off += align_up(sizeof(float) * (size_t)b * (h2 > 0 ? h2 : 1), 256);
    w.dvr_off = off; off += align_up(sizeof(float) * (size_t)b, 256);
    w.part_off = off; off += align_up(sizeof(float) * (size_t)(b > 0 ? b : 1) * (hp + 1), 256);
    w.total = off;
    return w;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__device__ __forceinline__ float block_sum_256(float v, float* s4) {
    // fixed-shape reduction (wave butterflies + 4 partials): deterministic
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = v;
    __syncthreads();
    return s4[0] + s4[1] + s4[2] + s4[3];
}

// block = 256 threads (4 waves), one graph per block
__global__ __launch_bounds__(256) void head_fwd_kernel(
    int H, int hp, int mode, const int* __restrict__ gptr, const float* __restrict__ h,
    const float* __restrict__ lin_w, const float* __restrict__ lin_b, const float* __restrict__ v0_w,
    const float* __restrict__ v0_b, const float* __restrict__ v1_w, const float* __restrict__ v1_b,
    float* __restrict__ q, float* __restrict__ out_v, float* __restrict__ adv_raw, float* __restrict__ pooled,
    int* __restrict__ amax, int* __restrict__ amin, float* __restrict__ z, float* __restrict__ vraw) {
    __shared__ __attribute__((aligned(16))) float s_w[128];
    __shared__ __attribute__((aligned(16))) float s_pool[4 * 128];
    __shared__ float s_z[64];
    __shared__ float s_red[4];
    __shared__ float s_v;
    __shared__ float s_mx[128], s_mn[128], s_sm[128];
    __shared__ int s_ax[128], s_an[128];
    const int g = blockIdx.x;
    const int r0 = gptr[g], r1 = gptr[g + 1], cnt = r1 - r0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H2 = H / 2, H4 = 4 * H;
    if (tid < 128) s_w[tid] = tid < H ? lin_w[tid] : 0.f;
    __syncthreads();

    // 1. advantages: four lanes per row (64 rows per pass); lane s of a row takes the 16-byte column groups s, s+4, ...
    //    (a wave-wide load = 16 rows x 64 contiguous bytes, up to eight of them in flight), then two butterfly adds.  One
    //    thread per row with a rolled column loop paid a memory round trip per 16 bytes: 47 us for a 171-row graph.
    const float lb = lin_b[0];
    float tsum = 0.f;
    const int sub = tid & 3, q4n = hp / 4;
    for (int row = r0 + (tid >> 2); row < r1; row += 64) {     // the four lanes of a row agree on the trip count
        const f32x4* hr = reinterpret_cast<const f32x4*>(h + (size_t)row * hp);
        const f32x4* wv = reinterpret_cast<const f32x4*>(s_w);
        f32x4 hv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = sub + 4 * u;
            hv[u] = c < q4n ? hr[c] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        float a = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = sub + 4 * u;
            if (c < q4n) {
                const f32x4 ww = wv[c];
                a += hv[u][0] * ww[0] + hv[u][1] * ww[1] + hv[u][2] * ww[2] + hv[u][3] * ww[3];
            }
        }
        a += __shfl_xor(a, 1);
        a += __shfl_xor(a, 2);
        if (sub == 0) {
            a += lb;
            adv_raw[row] = a;
            const float t = 2.f * tanhf(a);
            tsum += t;
            if (mode == 2) q[row] = t;
            if (mode >= 3) q[row] = a;          // raw advantages (HeadNetwork.forward)
        }
    }
    if (mode == 2 || mode == 4) return;
    const float adv_total = block_sum_256(tsum, s_red);

    // 2. pooling: column c = tid & 127, two row phases (even / odd rows), first-index ties
    {
        const int c = tid & 127, ph = tid >> 7;
        float sum = 0.f, mx = -INFINITY, mn = INFINITY;
        int ax = -1, an = -1;
        if (c < H) {
            // eight rows' loads in flight per step (the loop-carried sum / extremum chain is cheap; one load per iteration
            // made the loop cost a memory round trip per row: 47 us for a 171-row graph)
            int row = r0 + ph;
            for (; row + 30 < r1; row += 32) {      // sixteen rows' loads in flight (same order of the sums and comparisons)
                float v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = h[(size_t)(row + 2 * u) * hp + c];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    sum += v[u];
                    if (v[u] > mx) { mx = v[u]; ax = row + 2 * u; }
                    if (v[u] < mn) { mn = v[u]; an = row + 2 * u; }
                }
            }
            for (; row + 14 < r1; row += 16) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = h[(size_t)(row + 2 * u) * hp + c];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    sum += v[u];
                    if (v[u] > mx) { mx = v[u]; ax = row + 2 * u; }
                    if (v[u] < mn) { mn = v[u]; an = row + 2 * u; }
                }
            }
            for (; row < r1; row += 2) {
                const float v = h[(size_t)row * hp + c];
                sum += v;
                if (v > mx) { mx = v; ax = row; }
                if (v < mn) { mn = v; an = row; }
            }
        }
        if (ph == 1) { s_sm[c] = sum; s_mx[c] = mx; s_mn[c] = mn; s_ax[c] = ax; s_an[c] = an; }
        __syncthreads();
        if (ph == 0 && c < H) {
            sum += s_sm[c];
            const float mx1 = s_mx[c], mn1 = s_mn[c];
            const int ax1 = s_ax[c], an1 = s_an[c];
            if (ax1 >= 0 && (mx1 > mx || (mx1 == mx && ax1 < ax))) { mx = mx1; ax = ax1; }
            if (an1 >= 0 && (mn1 < mn || (mn1 == mn && an1 < an))) { mn = mn1; an = an1; }
            if (cnt == 0) { mx = 0.f; mn = 0.f; }
            const float mean = sum / (float)max(cnt, 1);
            s_pool[c] = sum; s_pool[H + c] = mx; s_pool[2 * H + c] = mn; s_pool[3 * H + c] = mean;
            float* pg = pooled + (size_t)g * H4;
            pg[c] = sum; pg[H + c] = mx; pg[2 * H + c] = mn; pg[3 * H + c] = mean;
            amax[(size_t)g * H + c] = ax;
            amin[(size_t)g * H + c] = an;
        }
    }
    __syncthreads();

    // 3. value MLP.  Thread (k = tid / 8, part = tid % 8) of a pass takes hidden unit k (32 per pass) and the 16-byte column
    //    groups part, part + 8, ... of its 4H-wide weight row: up to 16 independent loads in flight, an 8-lane butterfly
    //    to finish (one wave per unit with a shuffle reduction per unit paid a memory round trip per unit).
    for (int k0 = 0; k0 < H2; k0 += 32) {
        const int k = k0 + (tid >> 3), part = tid & 7;
        const f32x4* wrow = reinterpret_cast<const f32x4*>(v0_w + (size_t)(k < H2 ? k : 0) * H4);
        f32x4 wv[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int q = part + 8 * j;
            wv[j] = (k < H2 && q < H) ? wrow[q] : f32x4{0.f, 0.f, 0.f, 0.f};       // a 4H-float row = H float4 groups
        }
        f32x4 p4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int q = part + 8 * j;
            if (q < H) p4 += wv[j] * reinterpret_cast<const f32x4*>(s_pool)[q];
        }
        float p = (p4[0] + p4[1]) + (p4[2] + p4[3]);
        p += __shfl_xor(p, 1);
        p += __shfl_xor(p, 2);
        p += __shfl_xor(p, 4);
        if (part == 0 && k < H2) {
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
            s_v = mode == 3 ? v : tanhf(v);
        }
    }
    __syncthreads();

    // 4. dueling combine (each thread re-reads the adv_raw it wrote itself: lane 0 of a row's four lanes)
    const float mean_adv = adv_total / (float)max(cnt, 1);
    const float V = s_v;
    if ((mode == 1 || mode == 3) && tid == 0) out_v[g] = V;
    if (mode == 3) return;
    if (sub == 0) {
        for (int row = r0 + (tid >> 2); row < r1; row += 64) {
            const float t = 2.f * tanhf(adv_raw[row]);
            q[row] = (mode == 0 ? V : 0.f) + t - mean_adv;
        }
    }
}

__global__ __launch_bounds__(256) void head_bwd_kernel(
    int H, int hp, int mode, const int* __restrict__ gptr, const float* __restrict__ h,
    const float* __restrict__ lin_w, const float* __restrict__ v0_w, const float* __restrict__ v1_w,
    const float* __restrict__ adv_raw, const int* __restrict__ amax, const int* __restrict__ amin,
    const float* __restrict__ z, const float* __restrict__ vraw, const float* __restrict__ dq,
    const float* __restrict__ d_out_v, float* __restrict__ dh, float* __restrict__ dadv, float* __restrict__ dz,
    float* __restrict__ dvr, float* __restrict__ lin_part /*[b][hp+1]*/, int mask_dh) {
    __shared__ float s_dp[4 * 128];
    __shared__ float s_dz[64];
    __shared__ int s_ax[128], s_an[128];
    __shared__ float s_red[4];
    __shared__ float s_acc[129];
    __shared__ __attribute__((aligned(16))) float s_dar[1024];   // per-row advantage gradient of this graph (graphs beyond 1024 rows re-read global)
    const int g = blockIdx.x;
    const int r0 = gptr[g], r1 = gptr[g + 1], cnt = r1 - r0;
    const int tid = threadIdx.x;
    const int H2 = H / 2;
    __shared__ float s_lw[128];
    if (tid < 128) s_lw[tid] = tid < H ? lin_w[tid] : 0.f;      // visible after the barrier in front of the row loops

    float mean_dq = 0.f, inv_cnt = 1.f / (float)max(cnt, 1);
    const bool has_value = mode != 2 && mode != 4, raw = mode >= 3;
    if (has_value) {
        float ps = 0.f;
        for (int row = r0 + tid; row < r1; row += 256) ps += dq[row];
        const float sdq = block_sum_256(ps, s_red);
        mean_dq = raw ? 0.f : sdq * inv_cnt;
        if (tid < H) { s_ax[tid] = amax[(size_t)g * H + tid]; s_an[tid] = amin[(size_t)g * H + tid]; }
        const float dV = mode == 0 ? sdq : d_out_v[g];
        const float dv = raw ? dV : dV * sech2f(vraw[g]);
        if (tid == 0) dvr[g] = dv;
        if (tid < H2) {
            const float zz = z[(size_t)g * H2 + tid];
            const float d = zz > 0.f ? v1_w[tid] * dv : 0.f;
            s_dz[tid] = d;
            dz[(size_t)g * H2 + tid] = d;
        }
        __syncthreads();
        // d pooled = v0_w^T dz ([H2] x [H2][4H]) in 16-byte column groups x two k phases (as the fused backward's prologue does
        // with four): 28 independent 16-byte loads per thread instead of 110 4-byte ones (MIX: this kernel was 36 us)
        {
            const int cq = tid & 127, kg = tid >> 7;
            f32x4 p4 = f32x4{0.f, 0.f, 0.f, 0.f};
            if (cq < H) {           // H4 / 4 == H column groups
                const f32x4* wq = reinterpret_cast<const f32x4*>(v0_w) + cq;
#pragma unroll 8
                for (int k = kg; k < H2; k += 2) p4 += wq[(size_t)k * H] * s_dz[k];
            }
            f32x4* s_p4 = reinterpret_cast<f32x4*>(s_dar);      // (s_dar is written only after the next barrier)
            if (kg == 1 && cq < H) s_p4[cq] = p4;
            __syncthreads();
            if (kg == 0 && cq < H) {
                p4 += s_p4[cq];                                 // fixed order: deterministic
#pragma unroll
                for (int j = 0; j < 4; ++j) s_dp[4 * cq + j] = p4[j];
            }
            __syncthreads();
        }
    }
    // per-row advantage gradient: thread per row
    for (int row = r0 + tid; row < r1; row += 256) {
        const float dar = raw ? dq[row] : (dq[row] - mean_dq) * 2.f * sech2f(adv_raw[row]);
        dadv[row] = dar;
        if (row - r0 < 1024) s_dar[row - r0] = dar;
    }
    __syncthreads();
    // dh rows: four lanes per row (64 rows per pass), 16-byte stores; advantage-linear gradient below: column c = tid&127.
    // Per column ONE 16-byte LDS read (linear weight | pooled-sum + mean part | max part | min part) and one 8-byte read
    // (rows of the maximum / minimum) instead of seven 4-byte ones; the mask rows of a pass are requested before its
    // arithmetic (eight loads in flight instead of one round trip per column group).
    {
        __shared__ f32x4 s_c4[128];
        __shared__ int s_i2[128][2];
        if (tid < 128) {
            const int c = tid;
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            int ia = -1, ib = -1;
            if (c < H) {
                v[0] = s_lw[c];
                if (has_value) { v[1] = s_dp[c] + s_dp[3 * H + c] * inv_cnt; v[2] = s_dp[H + c]; v[3] = s_dp[2 * H + c]; ia = s_ax[c]; ib = s_an[c]; }
            }
            s_c4[c] = v;
            s_i2[c][0] = ia; s_i2[c][1] = ib;
        }
        __syncthreads();
        const int sub = tid & 3, q4n = hp / 4;
        for (int row = r0 + (tid >> 2); row < r1; row += 64) {
            const float dar = (row - r0 < 1024) ? s_dar[row - r0] : dadv[row];
            f32x4* dr = reinterpret_cast<f32x4*>(dh + (size_t)row * hp);
            f32x4 hv[8];
            if (mask_dh) {      // dh * [h > 0]: the gradient the ReLU layer underneath receives
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int q = sub + 4 * u;
                    hv[u] = q < q4n ? reinterpret_cast<const f32x4*>(h + (size_t)row * hp)[q] : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int q = sub + 4 * u;
                if (q < q4n) {
                    f32x4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int c = 4 * q + j;
                        const f32x4 k4 = s_c4[c];
                        float t = dar * k4[0];                   // (pad columns: all four constants are zero)
                        if (has_value) {
                            t += k4[1];
                            if (s_i2[c][0] == row) t += k4[2];
                            if (s_i2[c][1] == row) t += k4[3];
                        }
                        v[j] = t;
                    }
                    if (mask_dh) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = hv[u][j] > 0.f ? v[j] : 0.f;
                    }
                    dr[q] = v;
                }
            }
        }
    }
    {
        const int c = tid & 127, ph = tid >> 7;
        float acc = 0.f, accb = 0.f;
        int row = r0 + ph;
        for (; row + 30 < r1 && c < hp; row += 32) {        // sixteen rows' loads in flight per step, same summation order
            float v[16], d[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int rr = row + 2 * u;
                v[u] = h[(size_t)rr * hp + c];
                d[u] = (rr - r0 < 1024) ? s_dar[rr - r0] : dadv[rr];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) { acc += d[u] * v[u]; accb += d[u]; }
        }
        for (; row + 14 < r1 && c < hp; row += 16) {        // eight rows' loads in flight per step, same summation order
            float v[8], d[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int rr = row + 2 * u;
                v[u] = h[(size_t)rr * hp + c];
                d[u] = (rr - r0 < 1024) ? s_dar[rr - r0] : dadv[rr];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc += d[u] * v[u]; accb += d[u]; }
        }
        for (; row < r1; row += 2) {
            const float d = (row - r0 < 1024) ? s_dar[row - r0] : dadv[row];
            if (c < hp) acc += d * h[(size_t)row * hp + c];
            accb += d;
        }
        if (ph == 1) { s_acc[c] = acc; if (c == 0) s_acc[128] = accb; }
        __syncthreads();
        if (ph == 0) {
            if (c < hp) lin_part[(size_t)g * (hp + 1) + c] = acc + s_acc[c];
            if (c == 0) lin_part[(size_t)g * (hp + 1) + hp] = accb + s_acc[128];
        }
    }
}

// value-head parameter gradients.  d_v0_w[k][c] = sum_g dz[g][k]*pooled[g][c]: grid (ceil(4H/64), H/2), block 256 =
// 64 columns x 4 graph phases, fixed-shape combine (deterministic).  Block (0,0) also does the three small vectors.
__global__ __launch_bounds__(256) void head_value_wgrad_kernel(int b, int H, const float* __restrict__ dz,
                                                             const float* __restrict__ dvr,
                                                             const float* __restrict__ pooled,
                                                             const float* __restrict__ z, float* __restrict__ d_v0_w,
                                                             float* __restrict__ d_v0_b, float* __restrict__ d_v1_w,
                                                             float* __restrict__ d_v1_b, int hp,
                                                             const float* __restrict__ lin_part,
                                                             float* __restrict__ d_lin_w, float* __restrict__ d_lin_b) {
    __shared__ float red[3][64];
    const int H2 = H / 2, H4 = 4 * H;
    const int k = blockIdx.y;
    const int tid = threadIdx.x, cl = tid & 63, ph = tid >> 6;
    if (k == H2) {      // the extra row of workgroups: the advantage linear's gradient from the per-graph partials (one wave
                        // per column, same order as head_lin_grad_reduce_kernel)
        for (int c = blockIdx.x * 4 + ph; c <= H; c += 4 * gridDim.x) {
            const int src = c < H ? c : hp;
            float s = 0.f;
            for (int g = cl; g < b; g += 64) s += lin_part[(size_t)g * (hp + 1) + src];
            s = wave_sum(s);
            if (cl == 0) { if (c < H) d_lin_w[c] = s; else d_lin_b[0] = s; }
        }
        return;
    }
    const int c = blockIdx.x * 64 + cl;
    float s = 0.f;
    if (c < H4) {
#pragma unroll 8
        for (int g = ph; g < b; g += 4) s += dz[(size_t)g * H2 + k] * pooled[(size_t)g * H4 + c];
    }
    if (ph > 0) red[ph - 1][cl] = s;
    __syncthreads();
    if (ph == 0 && c < H4) d_v0_w[(size_t)k * H4 + c] = s + red[0][cl] + red[1][cl] + red[2][cl];
    if (blockIdx.x == 0) {
        // d_v0_b[k], d_v1_w[k] (wave 0 / wave 1), d_v1_b (block k == 0, wave 2)
        const int lane = tid & 63, wave = tid >> 6;
        float p = 0.f;
        if (wave == 0) { for (int g = lane; g < b; g += 64) p += dz[(size_t)g * H2 + k]; }
        else if (wave == 1) { for (int g = lane; g < b; g += 64) p += dvr[g] * z[(size_t)g * H2 + k]; }
        else if (wave == 2 && k == 0) { for (int g = lane; g < b; g += 64) p += dvr[g]; }
        p = wave_sum(p);
        if (lane == 0) {
            if (wave == 0) d_v0_b[k] = p;
            else if (wave == 1) d_v1_w[k] = p;
            else if (wave == 2 && k == 0) d_v1_b[0] = p;
        }
    }
}

// advantage Linear gradient: sum the per-graph partials; one wave per output column (hp+1 of them)
__global__ __launch_bounds__(64) void head_lin_grad_reduce_kernel(int b, int hp, int H, const float* __restrict__ part,
                                                                 float* __restrict__ d_lin_w, float* __restrict__ d_lin_b) {
    const int c = blockIdx.x, lane = threadIdx.x;   // c in [0, H]  (H == bias)
    const int src = c < H ? c : hp;
    float s = 0.f;
    for (int g = lane; g < b; g += 64) s += part[(size_t)g * (hp + 1) + src];
    s = wave_sum(s);
    if (lane == 0) { if (c < H) d_lin_w[c] = s; else d_lin_b[0] = s; }
}

// ---- head tail of the two_headed family (GN0/models.py:901-918): value_head_type="linear" over value_aggr_types=("mean",),
// i.e. value_g = Linear(H,1)(mean_{i in g} h_i) = mean_i(val_w . h_i) + val_b (HeadNetwork.forward, GN0/models.py:374-384),
// same advantage linear and the same dueling combine / modes as the MLP tail above.  One workgroup per graph, four lanes
// per row; saved = adv_raw [n] | vraw [b].
struct HeadLinSaved { size_t adv_off, v_off, total; };
static HeadLinSaved head_lin_saved_plan(int n, int b) {
    HeadLinSaved s;
    size_t off = 0;
    s.adv_off = off; off += align_up(sizeof(float) * (size_t)n, 256);
    s.v_off = off; off += align_up(sizeof(float) * (size_t)b, 256);
    s.total = off;
    return s;
}
struct HeadLinWs { size_t dvr_off, lpart_off, vpart_off, total; };
static HeadLinWs head_lin_ws_plan(int b, int hidden) {
    HeadLinWs w;
    const int hp = padded_width(hidden);
    size_t off = 0;
    w.dvr_off = off; off += align_up(sizeof(float) * (size_t)(b > 0 ? b : 1), 256);
    w.lpart_off = off; off += align_up(sizeof(float) * (size_t)(b > 0 ? b : 1) * (hp + 1), 256);
    w.vpart_off = off; off += align_up(sizeof(float) * (size_t)(b > 0 ? b : 1) * (hp + 1), 256);
    w.total = off;
    return w;
}

__global__ __launch_bounds__(256) void head_linear_fwd_kernel(
    int H, int hp, int mode, const int* __restrict__ gptr, const float* __restrict__ h, const float* __restrict__ lin_w,
    const float* __restrict__ lin_b, const float* __restrict__ val_w, const float* __restrict__ val_b,
    float* __restrict__ q, float* __restrict__ out_v, float* __restrict__ adv_raw, float* __restrict__ vraw) {
    __shared__ __attribute__((aligned(16))) float s_w[128], s_vw[128];
    __shared__ float s_red[4];
    const int g = blockIdx.x;
    const int r0 = gptr[g], r1 = gptr[g + 1], cnt = r1 - r0;
    const int tid = threadIdx.x;
    const bool has_value = mode != 2 && mode != 4;
    if (tid < 128) { s_w[tid] = tid < H ? lin_w[tid] : 0.f; s_vw[tid] = (has_value && tid < H) ? val_w[tid] : 0.f; }
    __syncthreads();
    const float lb = lin_b[0];
    float tsum = 0.f, usum = 0.f;
    const int sub = tid & 3, q4n = hp / 4;
    for (int row = r0 + (tid >> 2); row < r1; row += 64) {
        const f32x4* hr = reinterpret_cast<const f32x4*>(h + (size_t)row * hp);
        float a = 0.f, u = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = sub + 4 * k;
            if (c < q4n) {
                const f32x4 hv = hr[c], ww = reinterpret_cast<const f32x4*>(s_w)[c], vw = reinterpret_cast<const f32x4*>(s_vw)[c];
                a += hv[0] * ww[0] + hv[1] * ww[1] + hv[2] * ww[2] + hv[3] * ww[3];
                u += hv[0] * vw[0] + hv[1] * vw[1] + hv[2] * vw[2] + hv[3] * vw[3];
            }
        }
        a += __shfl_xor(a, 1); a += __shfl_xor(a, 2);
        u += __shfl_xor(u, 1); u += __shfl_xor(u, 2);
        if (sub == 0) {
            a += lb;
            adv_raw[row] = a;
            const float t = 2.f * tanhf(a);
            tsum += t;
            usum += u;
            if (mode == 2) q[row] = t;
            if (mode >= 3) q[row] = a;
        }
    }
    if (!has_value) return;
    const float adv_total = block_sum_256(tsum, s_red);
    const float u_total = block_sum_256(usum, s_red);
    const float v = u_total / (float)max(cnt, 1) + val_b[0];       // (an empty graph pools to zeros)
    if (tid == 0) vraw[g] = v;
    const float V = mode == 3 ? v : tanhf(v);
    if ((mode == 1 || mode == 3) && tid == 0) out_v[g] = V;
    if (mode == 3) return;
    const float mean_adv = adv_total / (float)max(cnt, 1);
    if (sub == 0) {
        for (int row = r0 + (tid >> 2); row < r1; row += 64) {
            const float t = 2.f * tanhf(adv_raw[row]);
            q[row] = (mode == 0 ? V : 0.f) + t - mean_adv;
        }
    }
}

// dh = dar * lin_w + (dv / cnt) * val_w;  per-graph partials lpart[g] = (sum_rows dar h | sum dar),
// vpart[g] = ((dv / cnt) sum_rows h | dv)
__global__ __launch_bounds__(256) void head_linear_bwd_kernel(
    int H, int hp, int mode, const int* __restrict__ gptr, const float* __restrict__ h, const float* __restrict__ lin_w,
    const float* __restrict__ val_w, const float* __restrict__ adv_raw, const float* __restrict__ vraw,
    const float* __restrict__ dq, const float* __restrict__ d_out_v, float* __restrict__ dh, float* __restrict__ dvr,
    float* __restrict__ lpart, float* __restrict__ vpart, int mask_dh) {
    __shared__ float s_lw[128], s_vw[128];
    __shared__ float s_red[4];
    __shared__ float s_acc[2][129];
    const int g = blockIdx.x;
    const int r0 = gptr[g], r1 = gptr[g + 1], cnt = r1 - r0;
    const int tid = threadIdx.x;
    const bool has_value = mode != 2 && mode != 4, raw = mode >= 3;
    if (tid < 128) { s_lw[tid] = tid < H ? lin_w[tid] : 0.f; s_vw[tid] = (has_value && tid < H) ? val_w[tid] : 0.f; }
    const float inv_cnt = 1.f / (float)max(cnt, 1);
    float mean_dq = 0.f, dvn = 0.f, dv = 0.f;
    if (has_value) {
        float ps = 0.f;
        for (int row = r0 + tid; row < r1; row += 256) ps += dq[row];
        const float sdq = block_sum_256(ps, s_red);
        mean_dq = raw ? 0.f : sdq * inv_cnt;
        const float dV = mode == 0 ? sdq : d_out_v[g];
        dv = raw ? dV : dV * sech2f(vraw[g]);
        if (tid == 0) dvr[g] = dv;
        dvn = cnt > 0 ? dv * inv_cnt : 0.f;
    }
    __syncthreads();
    {
        const int sub = tid & 3, q4n = hp / 4;
        for (int row = r0 + (tid >> 2); row < r1; row += 64) {
            const float dar = raw ? dq[row] : (dq[row] - mean_dq) * 2.f * sech2f(adv_raw[row]);
            f32x4* dr = reinterpret_cast<f32x4*>(dh + (size_t)row * hp);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int qq = sub + 4 * k;
                if (qq < q4n) {
                    f32x4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int c = 4 * qq + j;
                        v[j] = c < H ? dar * s_lw[c] + dvn * s_vw[c] : 0.f;
                    }
                    if (mask_dh) {
                        const f32x4 hv = reinterpret_cast<const f32x4*>(h + (size_t)row * hp)[qq];
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = hv[j] > 0.f ? v[j] : 0.f;
                    }
                    dr[qq] = v;
                }
            }
        }
    }
    {   // column c = tid & 127, two row phases, fixed order
        const int c = tid & 127, ph = tid >> 7;
        float acc = 0.f, accb = 0.f, hs = 0.f;
        for (int row = r0 + ph; row < r1; row += 2) {
            const float dar = raw ? dq[row] : (dq[row] - mean_dq) * 2.f * sech2f(adv_raw[row]);
            const float hv = c < hp ? h[(size_t)row * hp + c] : 0.f;
            acc += dar * hv;
            hs += hv;
            accb += dar;
        }
        if (ph == 1) { s_acc[0][c] = acc; s_acc[1][c] = hs; if (c == 0) s_acc[0][128] = accb; }
        __syncthreads();
        if (ph == 0) {
            if (c < hp) {
                lpart[(size_t)g * (hp + 1) + c] = acc + s_acc[0][c];
                vpart[(size_t)g * (hp + 1) + c] = dvn * (hs + s_acc[1][c]);
            }
            if (c == 0) {
                lpart[(size_t)g * (hp + 1) + hp] = accb + s_acc[0][128];
                vpart[(size_t)g * (hp + 1) + hp] = dv;
            }
        }
    }
}

// ---- HexAra policy head pieces (GN0/torch_script_models.py:286-379) ----------------------------------------------------------
// (a) the policy head's LAST layer is a SAGEConv(H, 1) (ModifiedBaseNet with out_channels=1, lines 123-144, 296):
//       out_i = b + w_r . h_i + mean_{j in N(i)} w_l . h_j
//     two dot products per row, then the mean of a scalar over the CSR: HBM-bound, no MFMA.
__global__ __launch_bounds__(256) void sage_scalar_dots_kernel(int n, int H, int hp, const float* __restrict__ h,
                                                              const float* __restrict__ wl, const float* __restrict__ wr,
                                                              float* __restrict__ s /*[n][2]*/) {
    __shared__ __attribute__((aligned(16))) float s_l[128], s_r[128];
    const int tid = threadIdx.x;
    if (tid < 128) { s_l[tid] = tid < H ? wl[tid] : 0.f; s_r[tid] = tid < H ? wr[tid] : 0.f; }
    __syncthreads();
    const int sub = tid & 3, q4n = hp / 4;
    const int row = blockIdx.x * 64 + (tid >> 2);
    if (row >= n) return;                                   // (the four lanes of a row leave together)
    const f32x4* hr = reinterpret_cast<const f32x4*>(h + (size_t)row * hp);
    float a = 0.f, u = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = sub + 4 * k;
        if (c < q4n) {
            const f32x4 hv = hr[c], lw = reinterpret_cast<const f32x4*>(s_l)[c], rw = reinterpret_cast<const f32x4*>(s_r)[c];
            a += hv[0] * lw[0] + hv[1] * lw[1] + hv[2] * lw[2] + hv[3] * lw[3];
            u += hv[0] * rw[0] + hv[1] * rw[1] + hv[2] * rw[2] + hv[3] * rw[3];
        }
    }
    a += __shfl_xor(a, 1); a += __shfl_xor(a, 2);
    u += __shfl_xor(u, 1); u += __shfl_xor(u, 2);
    if (sub == 0) { s[2 * (size_t)row] = a; s[2 * (size_t)row + 1] = u; }
}

__global__ __launch_bounds__(256) void sage_scalar_gather_kernel(int n, const int* __restrict__ rowptr,
                                                                const int* __restrict__ col, const float* __restrict__ invdeg,
                                                                const float* __restrict__ s, const float* __restrict__ bias,
                                                                float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float acc = 0.f;
    for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) acc += s[2 * (size_t)col[e]];       // ascending neighbour order
    out[i] = bias[0] + s[2 * (size_t)i + 1] + invdeg[i] * acc;
}

// backward: ds_r = dout,  ds_l[j] = sum_{i in T(j)} dout_i / deg_i,  dh_j = ds_l[j] w_l + ds_r[j] w_r;
// block partials (64 rows): lpart[blk] = (sum ds_l h | sum dout), rpart[blk] = (sum ds_r h | 0)
__global__ __launch_bounds__(256) void sage_scalar_bwd_kernel(int n, int H, int hp, const int* __restrict__ rowptr_t,
                                                             const int* __restrict__ col_t, const float* __restrict__ invdeg,
                                                             const float* __restrict__ h, const float* __restrict__ wl,
                                                             const float* __restrict__ wr, const float* __restrict__ dout,
                                                             float* __restrict__ dh, float* __restrict__ lpart,
                                                             float* __restrict__ rpart) {
    __shared__ float s_l[128], s_r[128];
    __shared__ float s_dl[64], s_dr[64];
    __shared__ float s_acc[2][129];
    const int tid = threadIdx.x;
    const int r0 = blockIdx.x * 64, r1 = min(n, r0 + 64);
    if (tid < 128) { s_l[tid] = tid < H ? wl[tid] : 0.f; s_r[tid] = tid < H ? wr[tid] : 0.f; }
    if (tid < 64) {
        const int j = r0 + tid;
        float dl = 0.f, dr = 0.f;
        if (j < n) {
            dr = dout[j];
            for (int e = rowptr_t[j]; e < rowptr_t[j + 1]; ++e) { const int i = col_t[e]; dl += dout[i] * invdeg[i]; }
        }
        s_dl[tid] = dl;
        s_dr[tid] = dr;
    }
    __syncthreads();
    {
        const int sub = tid & 3, q4n = hp / 4, row = r0 + (tid >> 2);
        if (row < n) {
            const float dl = s_dl[tid >> 2], dr = s_dr[tid >> 2];
            f32x4* d = reinterpret_cast<f32x4*>(dh + (size_t)row * hp);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int q = sub + 4 * k;
                if (q < q4n) {
                    f32x4 v;
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) { const int c = 4 * q + jj; v[jj] = c < H ? dl * s_l[c] + dr * s_r[c] : 0.f; }
                    d[q] = v;
                }
            }
        }
    }
    {
        const int c = tid & 127, ph = tid >> 7;
        float al = 0.f, ar = 0.f, ab = 0.f;
        for (int row = r0 + ph; row < r1; row += 2) {
            const float hv = c < hp ? h[(size_t)row * hp + c] : 0.f;
            al += s_dl[row - r0] * hv;
            ar += s_dr[row - r0] * hv;
            ab += s_dr[row - r0];
        }
        if (ph == 1) { s_acc[0][c] = al; s_acc[1][c] = ar; if (c == 0) s_acc[0][128] = ab; }
        __syncthreads();
        if (ph == 0) {
            if (c < hp) {
                lpart[(size_t)blockIdx.x * (hp + 1) + c] = al + s_acc[0][c];
                rpart[(size_t)blockIdx.x * (hp + 1) + c] = ar + s_acc[1][c];
            }
            if (c == 0) {
                lpart[(size_t)blockIdx.x * (hp + 1) + hp] = ab + s_acc[0][128];
                rpart[(size_t)blockIdx.x * (hp + 1) + hp] = 0.f;
            }
        }
    }
}

// (b) output surgery + scatter_log_softmax (lines 326-378): per graph g the output segment holds the logits of its
//     non-terminal nodes (rows gptr[g]+2 ..) and, when swapping is allowed in g, the graph's swap logit behind them;
//     segment start = gptr[g] - 2g + (number of swap slots of graphs < g) = output_batch_ptr[g]; log-softmax per segment.
//     swap flag of graph g (lines 337-347): feature 2 of the graph's LAST node for g < b-1, of its FIRST node for g = b-1.
__device__ __forceinline__ int swap_flag(int g, int b, const int* gptr, const float* x, int xs, int swap_allowed) {
    if (!swap_allowed) return 0;
    const int row = g < b - 1 ? gptr[g + 1] - 1 : gptr[g];
    return x[(size_t)row * xs + 2] != 0.f ? 1 : 0;
}
__device__ __forceinline__ float block_max_256(float v, float* s4) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(s4[0], s4[1]), fmaxf(s4[2], s4[3]));
}
__device__ __forceinline__ int block_isum_256(int v, int* s4) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = v;
    __syncthreads();
    return s4[0] + s4[1] + s4[2] + s4[3];
}

__global__ __launch_bounds__(256) void policy_lsm_fwd_kernel(int b, const int* __restrict__ gptr, const float* __restrict__ x,
                                                            int xs, int swap_allowed, const float* __restrict__ pi_raw,
                                                            const float* __restrict__ should_swap, float* __restrict__ out_pi,
                                                            int64_t* __restrict__ out_gi, int64_t* __restrict__ out_ptr) {
    __shared__ float s4[4];
    __shared__ int i4[4];
    const int g = blockIdx.x, tid = threadIdx.x;
    int before = 0;
    for (int j = tid; j < g; j += 256) before += swap_flag(j, b, gptr, x, xs, swap_allowed);
    before = block_isum_256(before, i4);
    const int flag = swap_flag(g, b, gptr, x, xs, swap_allowed);
    const int r0 = gptr[g] + 2, r1 = gptr[g + 1];
    const int len = max(r1 - r0, 0) + flag;
    const int64_t o0 = (int64_t)gptr[g] - 2 * (int64_t)g + before;
    auto val = [&](int k) { return k < r1 - r0 ? pi_raw[r0 + k] : should_swap[g]; };
    float mx = -INFINITY;
    for (int k = tid; k < len; k += 256) mx = fmaxf(mx, val(k));
    mx = block_max_256(mx, s4);
    float se = 0.f;
    for (int k = tid; k < len; k += 256) se += expf(val(k) - mx);
    se = block_sum_256(se, s4);
    const float lse = logf(se);
    for (int k = tid; k < len; k += 256) {
        out_pi[o0 + k] = val(k) - mx - lse;
        out_gi[o0 + k] = g;
    }
    if (tid == 0) {
        out_ptr[g] = o0;
        if (g == b - 1) out_ptr[b] = o0 + len;
    }
}

// d logit_k = d out_k - softmax_k * sum(d out);  terminal rows get 0
__global__ __launch_bounds__(256) void policy_lsm_bwd_kernel(int b, const int* __restrict__ gptr, const float* __restrict__ x,
                                                            int xs, int swap_allowed, const int64_t* __restrict__ out_ptr,
                                                            const float* __restrict__ out_pi, const float* __restrict__ d_out,
                                                            float* __restrict__ d_pi_raw, float* __restrict__ d_should_swap) {
    __shared__ float s4[4];
    const int g = blockIdx.x, tid = threadIdx.x;
    const int flag = swap_flag(g, b, gptr, x, xs, swap_allowed);
    const int r0 = gptr[g] + 2, r1 = gptr[g + 1];
    const int nn = max(r1 - r0, 0), len = nn + flag;
    const int64_t o0 = out_ptr[g];
    float sd = 0.f;
    for (int k = tid; k < len; k += 256) sd += d_out[o0 + k];
    sd = block_sum_256(sd, s4);
    for (int k = tid; k < len; k += 256) {
        const float d = d_out[o0 + k] - expf(out_pi[o0 + k]) * sd;
        if (k < nn) d_pi_raw[r0 + k] = d;
        else d_should_swap[g] = d;
    }
    if (tid < 2 && gptr[g] + tid < r1) d_pi_raw[gptr[g] + tid] = 0.f;
    if (tid == 0 && !flag && d_should_swap) d_should_swap[g] = 0.f;
}

int launch_head_param_grads(int b, int hidden, int mode, const float* dz, const float* dvr, const float* pooled,
                            const float* z, const float* lin_part, float* d_lin_w, float* d_lin_b, float* d_v0_w,
                            float* d_v0_b, float* d_v1_w, float* d_v1_b, hipStream_t st) {
    const int hp = padded_width_wide(hidden), H2 = hidden / 2, H4 = 4 * hidden;
    if (mode != 2 && mode != 4 && H2 > 0)       // one launch: the value MLP's gradients + (last row of workgroups) the linear's
        head_value_wgrad_kernel<<<dim3((H4 + 63) / 64, H2 + 1), 256, 0, st>>>(b, hidden, dz, dvr, pooled, z, d_v0_w, d_v0_b,
                                                                              d_v1_w, d_v1_b, hp, lin_part, d_lin_w, d_lin_b);
    else
        head_lin_grad_reduce_kernel<<<hidden + 1, 64, 0, st>>>(b, hp, hidden, lin_part, d_lin_w, d_lin_b);
    return HEXGNN_OK;
}


// ---- TD loss on the selected nodes: loss = mean_j w_j * l(q[sel_j] - tgt_j), l = d^2 ("mse") or Huber(delta 1) -----------
// The reference's training loop gathers Q(s, a) with torch indexing and calls the loss in torch (Rainbow agent,
// --loss_fn=mse, importance weights of the prioritized replay): ~15 tiny kernels forward + backward (gather, sub, pow,
// mean, sort-based index_put ...).  Here: one single-workgroup kernel forward (fixed-shape tree => deterministic) and
// one memset + one scatter kernel backward.  td[j] = q[sel_j] - tgt_j is returned for the priority update.
__global__ __launch_bounds__(256) void td_loss_fwd_kernel(int n, int k, const float* __restrict__ q,
                                                         const int64_t* __restrict__ sel, const float* __restrict__ tgt,
                                                         const float* __restrict__ w, int loss_fn,
                                                         float* __restrict__ loss, float* __restrict__ td) {
    __shared__ float red[256];
    float acc = 0.f;
    for (int j = threadIdx.x; j < k; j += 256) {
        const int64_t i = sel[j];
        const float d = (i >= 0 && i < n) ? q[i] - tgt[j] : 0.f;
        td[j] = d;
        const float a = fabsf(d);
        const float l = loss_fn == 0 ? d * d : (a <= 1.f ? 0.5f * d * d : a - 0.5f);
        acc += (w ? w[j] : 1.f) * l;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = red[0] / (float)(k > 0 ? k : 1);
}

// One launch: every workgroup clears its 1024-entry range of dq, then accumulates the selected nodes that fall into it
// (the zeroing memset used to be a launch of its own).  Bit-reproducible with duplicated selections (PER samples with
// replacement): an LDS counter per node of the range says how many entries of the current 1024-entry chunk name it (integer
// atomics: order-free).  A node named once in the chunk gets one add (chunks are barrier-separated, so its adds arrive in
// chunk order); a node named twice or more is summed in list order by its first entry and added once.  (Until round 3 two
// entries went through two float atomics: order-free only onto a ZERO word, i.e. wrong from the second chunk on -- lists
// above 1024 entries were not bit-reproducible; found by the 2300-entry test of the one-launch form.)
__global__ __launch_bounds__(256) void td_loss_bwd_kernel(int n, int k, const int64_t* __restrict__ sel,
                                                         const float* __restrict__ td, const float* __restrict__ w,
                                                         int loss_fn, const float* __restrict__ gloss,
                                                         float* __restrict__ dq) {
    __shared__ __attribute__((aligned(16))) int s_i[1024];
    __shared__ __attribute__((aligned(16))) float s_g[1024];
    __shared__ int s_cnt[1024];
    const int lo = blockIdx.x * 1024, hi = min(lo + 1024, n);
    for (int i = lo + threadIdx.x; i < hi; i += 256) dq[i] = 0.f;
    const float gl = gloss[0] / (float)k;
    for (int c0 = 0; c0 < k; c0 += 1024) {
        const int kk = min(1024, k - c0);
        __syncthreads();
        for (int j = threadIdx.x; j < 1024; j += 256) { s_cnt[j] = 0; s_i[j] = -1; s_g[j] = 0.f; }
        __syncthreads();
        for (int j = threadIdx.x; j < kk; j += 256) {
            const int64_t i = sel[c0 + j];
            const bool mine = i >= lo && i < hi;
            const float d = td[c0 + j];
            const float dl = loss_fn == 0 ? 2.f * d : fminf(fmaxf(d, -1.f), 1.f);
            s_i[j] = mine ? (int)i : -1;
            s_g[j] = gl * (w ? w[c0 + j] : 1.f) * dl;
            if (mine) atomicAdd(&s_cnt[(int)i - lo], 1);
        }
        __syncthreads();
        for (int j = threadIdx.x; j < kk; j += 256) {
            const int i = s_i[j];
            if (i < 0) continue;                       // not in this workgroup's range (most entries)
            if (s_cnt[i - lo] == 1) { atomicAdd(dq + i, s_g[j]); continue; }
            // three or more: is an earlier entry naming the same node? sum of the later ones, in list order
            bool first = true;
            float acc = s_g[j];
            for (int q4 = 0; q4 < (kk + 3) / 4; ++q4) {
                const int4 ii = reinterpret_cast<const int4*>(s_i)[q4];
                const f32x4 gg = reinterpret_cast<const f32x4*>(s_g)[q4];
                const int q = 4 * q4;
                first = first && !((ii.x == i && q < j) || (ii.y == i && q + 1 < j) || (ii.z == i && q + 2 < j) ||
                                   (ii.w == i && q + 3 < j));
                acc += (ii.x == i && q > j) ? gg[0] : 0.f;
                acc += (ii.y == i && q + 1 > j) ? gg[1] : 0.f;
                acc += (ii.z == i && q + 2 > j) ? gg[2] : 0.f;
                acc += (ii.w == i && q + 3 > j) ? gg[3] : 0.f;
            }
            if (first) atomicAdd(dq + i, acc);   // one add per node and chunk (chunks of 1024 entries are barrier-separated)
        }
    }
}


// Forward AND backward of the TD loss in ONE launch (the DQN update differentiates the loss itself, so d loss / d q is
// known the moment the loss is: grad_loss == 1): every workgroup owns a 1024-node range of dq exactly as
// td_loss_bwd_kernel does, with td_j = q[sel_j] - target_j formed on the fly; workgroup 0 additionally writes td[] and the
// mean.  Same fixed-shape sums / commuting atomics as the two-launch form: bit-identical results.
__global__ __launch_bounds__(256) void td_loss_fused_kernel(int n, int k, const float* __restrict__ q,
                                                           const int64_t* __restrict__ sel, const float* __restrict__ tgt,
                                                           const float* __restrict__ w, int loss_fn,
                                                           float* __restrict__ loss, float* __restrict__ td,
                                                           float* __restrict__ dq) {
    __shared__ __attribute__((aligned(16))) int s_i[1024];
    __shared__ __attribute__((aligned(16))) float s_g[1024];
    __shared__ int s_cnt[1024];
    __shared__ float red[256];
    const int lo = blockIdx.x * 1024, hi = min(lo + 1024, n);
    for (int i = lo + threadIdx.x; i < hi; i += 256) dq[i] = 0.f;
    const float gl = 1.f / (float)(k > 0 ? k : 1);
    float acc_loss = 0.f;
    for (int c0 = 0; c0 < k; c0 += 1024) {
        const int kk = min(1024, k - c0);
        __syncthreads();
        for (int j = threadIdx.x; j < 1024; j += 256) { s_cnt[j] = 0; s_i[j] = -1; s_g[j] = 0.f; }
        __syncthreads();
        for (int j = threadIdx.x; j < kk; j += 256) {
            const int64_t i = sel[c0 + j];
            const bool inside = i >= 0 && i < n;
            const bool mine = i >= lo && i < hi;
            const float d = inside ? q[i] - tgt[c0 + j] : 0.f;
            const float wj = w ? w[c0 + j] : 1.f;
            if (blockIdx.x == 0) {
                td[c0 + j] = d;
                const float a = fabsf(d);
                acc_loss += wj * (loss_fn == 0 ? d * d : (a <= 1.f ? 0.5f * d * d : a - 0.5f));
            }
            const float dl = loss_fn == 0 ? 2.f * d : fminf(fmaxf(d, -1.f), 1.f);
            s_i[j] = mine ? (int)i : -1;
            s_g[j] = gl * wj * dl;
            if (mine) atomicAdd(&s_cnt[(int)i - lo], 1);
        }
        __syncthreads();
        for (int j = threadIdx.x; j < kk; j += 256) {
            const int i = s_i[j];
            if (i < 0) continue;
            if (s_cnt[i - lo] == 1) { atomicAdd(dq + i, s_g[j]); continue; }
            bool first = true;
            float acc = s_g[j];
            for (int q4 = 0; q4 < (kk + 3) / 4; ++q4) {
                const int4 ii = reinterpret_cast<const int4*>(s_i)[q4];
                const f32x4 gg = reinterpret_cast<const f32x4*>(s_g)[q4];
                const int qq = 4 * q4;
                first = first && !((ii.x == i && qq < j) || (ii.y == i && qq + 1 < j) || (ii.z == i && qq + 2 < j) ||
                                   (ii.w == i && qq + 3 < j));
                acc += (ii.x == i && qq > j) ? gg[0] : 0.f;
                acc += (ii.y == i && qq + 1 > j) ? gg[1] : 0.f;
                acc += (ii.z == i && qq + 2 > j) ? gg[2] : 0.f;
                acc += (ii.w == i && qq + 3 > j) ? gg[3] : 0.f;
            }
            if (first) atomicAdd(dq + i, acc);
        }
    }
    if (blockIdx.x == 0) {      // the mean, in td_loss_fwd_kernel's reduction shape (thread t sums entries t, t+256, ...)
        red[threadIdx.x] = acc_loss;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) loss[0] = red[0] / (float)(k > 0 ? k : 1);
    }
}


// ---- acting: per-graph epsilon-greedy / argmax over the non-terminal nodes, mapped to vertex ids ----------------------
// One wave per graph.  Greedy = first node attaining the maximum of q[gptr[g]+2 : gptr[g+1]] (torch.argmax tie rule;
// GN0/RainbowDQN/evaluate_elo.py:253-266); with u given, env g explores when u[2g] < eps and then plays node
// 2 + floor(u[2g+1] * (n_g - 2)).  Outputs the node rank inside its graph, the vertex id (backmap, what
// Env_manager.validate_actions returns, multi_env_manager.py:62-64) and the exploratory flag.
__global__ __launch_bounds__(64) void select_actions_kernel(int b, const int* __restrict__ gptr, const float* __restrict__ q,
                                                          const int64_t* __restrict__ backmap, float eps,
                                                          const float* __restrict__ u, int* __restrict__ act_vertex,
                                                          int* __restrict__ act_rank, unsigned char* __restrict__ expl) {
    const int g = blockIdx.x, lane = threadIdx.x;
    const int r0 = gptr[g], r1 = gptr[g + 1];
    float best = -INFINITY;
    int arg = 0x7fffffff;
    for (int i = r0 + 2 + lane; i < r1; i += 64) {
        const float v = q[i];
        if (v > best || (v == best && i < arg)) { best = v; arg = i; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ob = __shfl_xor(best, off);
        const int oa = __shfl_xor(arg, off);
        if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
    }
    if (lane == 0) {
        int rank = arg == 0x7fffffff ? -1 : arg - r0;      // -1: graph without a legal move
        unsigned char ex = 0;
        const int nact = r1 - r0 - 2;
        if (u && nact > 0 && u[2 * g] < eps) {
            int k = (int)(u[2 * g + 1] * (float)nact);
            if (k >= nact) k = nact - 1;
            rank = 2 + k;
            ex = 1;
        }
        act_rank[g] = rank;
        if (act_vertex) act_vertex[g] = rank >= 0 ? (backmap ? (int)backmap[r0 + rank] : rank) : -1;
        if (expl) expl[g] = ex;
    }
}

}  // namespace hexgnn

using namespace hexgnn;

extern "C" {

int hexgnn_select_actions(int b, const int* gptr, const float* q, const int64_t* backmap, float eps, const float* u,
                          int* action_vertex, int* action_rank, uint8_t* exploratory, hexgnn_stream_t stream_) {
    if (b < 0 || (b > 0 && (!gptr || !q || !action_rank))) return HEXGNN_EINVAL;
    if (b == 0) return HEXGNN_OK;
    select_actions_kernel<<<b, 64, 0, (hipStream_t)stream_>>>(b, gptr, q, backmap, eps, u, action_vertex, action_rank,
                                                              exploratory);
    return check_launch();
}

int hexgnn_td_loss_forward(int n, int k, const float* q, const int64_t* sel, const float* target, const float* weights,
                           int loss_fn, float* loss, float* td, hexgnn_stream_t stream_) {
    if (n < 0 || k < 0 || loss_fn < 0 || loss_fn > 1 || !loss || (k > 0 && (!q || !sel || !target || !td))) return HEXGNN_EINVAL;
    td_loss_fwd_kernel<<<1, 256, 0, (hipStream_t)stream_>>>(n, k, q, sel, target, weights, loss_fn, loss, td);
    return check_launch();
}

int hexgnn_td_loss_backward(int n, int k, const int64_t* sel, const float* td, const float* weights, int loss_fn,
                            const float* grad_loss, float* dq, hexgnn_stream_t stream_) {
    hipStream_t st = (hipStream_t)stream_;
    if (n < 0 || k < 0 || loss_fn < 0 || loss_fn > 1 || !grad_loss || (n > 0 && !dq) || (k > 0 && (!sel || !td))) return HEXGNN_EINVAL;
    if (n > 0) td_loss_bwd_kernel<<<(n + 1023) / 1024, 256, 0, st>>>(n, k, sel, td, weights, loss_fn, grad_loss, dq);
    return check_launch();
}

int hexgnn_td_loss_forward_backward(int n, int k, const float* q, const int64_t* sel, const float* target,
                                    const float* weights, int loss_fn, float* loss, float* td, float* dq,
                                    hexgnn_stream_t stream_) {
    if (n < 0 || k < 0 || loss_fn < 0 || loss_fn > 1 || !loss || (n > 0 && !dq) || (k > 0 && (!q || !sel || !target || !td)))
        return HEXGNN_EINVAL;
    td_loss_fused_kernel<<<n > 0 ? (n + 1023) / 1024 : 1, 256, 0, (hipStream_t)stream_>>>(n, k, q, sel, target, weights,
                                                                                        loss_fn, loss, td, dq);
    return check_launch();
}

size_t hexgnn_head_saved_bytes(int n, int b, int hidden) {
    if (n < 0 || b < 0 || padded_width_wide(hidden) < 0) return 0;
    return head_saved_plan(n, b, hidden).total;
}

int hexgnn_head_forward(int n, int b, int hidden, int mode, const int* gptr, const float* h, const float* lin_w,
                        const float* lin_b, const float* v0_w, const float* v0_b, const float* v1_w,
                        const float* v1_b, float* q, float* out_v, void* saved, hexgnn_stream_t stream_) {
    const int hp = padded_width_wide(hidden);
    if (hp < 0 || hidden < 2) return HEXGNN_EUNSUPPORTED;
    if (n < 0 || b < 0 || mode < 0 || mode > 4) return HEXGNN_EINVAL;
    if (!gptr || !lin_w || !lin_b || !saved) return HEXGNN_EINVAL;
    if (mode != 2 && mode != 4 && (!v0_w || !v0_b || !v1_w || !v1_b)) return HEXGNN_EINVAL;
    if ((mode == 1 || mode == 3) && !out_v) return HEXGNN_EINVAL;
    if (n > 0 && (!h || !q)) return HEXGNN_EINVAL;
    if (b == 0) return HEXGNN_OK;
    if (hidden > 16 * kMaxNT)
        return wide_head_forward(n, b, hidden, mode, gptr, h, lin_w, lin_b, v0_w, v0_b, v1_w, v1_b, q, out_v, saved,
                                 (hipStream_t)stream_);
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
    if (n < 0 || b < 0 || padded_width_wide(hidden) < 0) return 0;
    return head_ws_plan(n, b, hidden).total;
}

int hexgnn_head_backward(int n, int b, int hidden, int mode, const int* gptr, const float* h, const float* lin_w,
                         const float* v0_w, const float* v1_w, const void* saved, const float* dq,
                         const float* d_out_v, float* dh, float* d_lin_w, float* d_lin_b, float* d_v0_w,
                         float* d_v0_b, float* d_v1_w, float* d_v1_b, void* workspace, size_t workspace_bytes,
                         hexgnn_stream_t stream_) {
    hipStream_t st = (hipStream_t)stream_;
    const int hp = padded_width_wide(hidden);
    if (hp < 0 || hidden < 2) return HEXGNN_EUNSUPPORTED;
    const int mode_in = mode;
    const int mask_dh = (mode & HEXGNN_HEAD_MASK_DH) ? 1 : 0;
    mode &= ~HEXGNN_HEAD_MASK_DH;
    if (n < 0 || b < 0 || mode < 0 || mode > 4) return HEXGNN_EINVAL;
    if (!gptr || !lin_w || !saved || !d_lin_w || !d_lin_b) return HEXGNN_EINVAL;
    if (mode != 2 && mode != 4 && (!v0_w || !v1_w || !d_v0_w || !d_v0_b || !d_v1_w || !d_v1_b)) return HEXGNN_EINVAL;
    if ((mode == 1 || mode == 3) && !d_out_v) return HEXGNN_EINVAL;
    if (n > 0 && (!h || !dq || !dh)) return HEXGNN_EINVAL;
    if (hidden > 16 * kMaxNT)
        return wide_head_backward(n, b, hidden, mode_in, gptr, h, lin_w, v0_w, v1_w, saved, dq, d_out_v, dh, d_lin_w, d_lin_b,
                                  d_v0_w, d_v0_b, d_v1_w, d_v1_b, workspace, workspace_bytes, st);
    const HeadWs w = head_ws_plan(n, b, hidden);
    if (!workspace || workspace_bytes < w.total) return HEXGNN_EWORKSPACE;
    const HeadSaved s = head_saved_plan(n, b, hidden);
    const char* sv = (const char*)saved;
    char* ws = (char*)workspace;
    float* dadv = (float*)(ws + w.dadv_off);
    float* dz = (float*)(ws + w.dz_off);
    float* dvr = (float*)(ws + w.dvr_off);
    float* part = (float*)(ws + w.part_off);
    KernelTimer kt(HEXGNN_K_HEAD_BWD, st);
    if (b > 0)
        head_bwd_kernel<<<b, 256, 0, st>>>(hidden, hp, mode, gptr, h, lin_w, v0_w, v1_w, (const float*)(sv + s.adv_off),
                                           (const int*)(sv + s.amax_off), (const int*)(sv + s.amin_off),
                                           (const float*)(sv + s.z_off), (const float*)(sv + s.v_off), dq, d_out_v,
                                           dh, dadv, dz, dvr, part, mask_dh);
    launch_head_param_grads(b, hidden, mode, dz, dvr, (const float*)(sv + s.pooled_off), (const float*)(sv + s.z_off),
                            part, d_lin_w, d_lin_b, d_v0_w, d_v0_b, d_v1_w, d_v1_b, st);
    return check_launch();
}

size_t hexgnn_head_linear_saved_bytes(int n, int b) {
    if (n < 0 || b < 0) return 0;
    return head_lin_saved_plan(n, b).total;
}

int hexgnn_head_linear_forward(int n, int b, int hidden, int mode, const int* gptr, const float* h, const float* lin_w,
                               const float* lin_b, const float* val_w, const float* val_b, float* q, float* out_v,
                               void* saved, hexgnn_stream_t stream_) {
    const int hp = padded_width(hidden);
    if (hp < 0) return HEXGNN_EUNSUPPORTED;
    if (n < 0 || b < 0 || mode < 0 || mode > 4) return HEXGNN_EINVAL;
    if (!gptr || !lin_w || !lin_b || !saved) return HEXGNN_EINVAL;
    if (mode != 2 && mode != 4 && (!val_w || !val_b)) return HEXGNN_EINVAL;
    if ((mode == 1 || mode == 3) && !out_v) return HEXGNN_EINVAL;
    if (n > 0 && (!h || !q)) return HEXGNN_EINVAL;
    if (b == 0) return HEXGNN_OK;
    const HeadLinSaved s = head_lin_saved_plan(n, b);
    char* sv = (char*)saved;
    head_linear_fwd_kernel<<<b, 256, 0, (hipStream_t)stream_>>>(hidden, hp, mode, gptr, h, lin_w, lin_b, val_w, val_b, q, out_v,
                                                               (float*)(sv + s.adv_off), (float*)(sv + s.v_off));
    return check_launch();
}

size_t hexgnn_head_linear_backward_workspace_bytes(int n, int b, int hidden) {
    if (n < 0 || b < 0 || padded_width(hidden) < 0) return 0;
    return head_lin_ws_plan(b, hidden).total;
}

int hexgnn_head_linear_backward(int n, int b, int hidden, int mode, const int* gptr, const float* h, const float* lin_w,
                                const float* val_w, const void* saved, const float* dq, const float* d_out_v, float* dh,
                                float* d_lin_w, float* d_lin_b, float* d_val_w, float* d_val_b, void* workspace,
                                size_t workspace_bytes, hexgnn_stream_t stream_) {
    hipStream_t st = (hipStream_t)stream_;
    const int hp = padded_width(hidden);
    if (hp < 0) return HEXGNN_EUNSUPPORTED;
    const int mask_dh = (mode & HEXGNN_HEAD_MASK_DH) ? 1 : 0;
    mode &= ~HEXGNN_HEAD_MASK_DH;
    if (n < 0 || b < 0 || mode < 0 || mode > 4) return HEXGNN_EINVAL;
    if (!gptr || !lin_w || !saved || !d_lin_w || !d_lin_b) return HEXGNN_EINVAL;
    const bool has_value = mode != 2 && mode != 4;
    if (has_value && (!val_w || !d_val_w || !d_val_b)) return HEXGNN_EINVAL;
    if ((mode == 1 || mode == 3) && !d_out_v) return HEXGNN_EINVAL;
    if (n > 0 && (!h || !dq || !dh)) return HEXGNN_EINVAL;
    const HeadLinWs w = head_lin_ws_plan(b, hidden);
    if (!workspace || workspace_bytes < w.total) return HEXGNN_EWORKSPACE;
    const HeadLinSaved s = head_lin_saved_plan(n, b);
    const char* sv = (const char*)saved;
    char* ws = (char*)workspace;
    float* lpart = (float*)(ws + w.lpart_off);
    float* vpart = (float*)(ws + w.vpart_off);
    if (b > 0)
        head_linear_bwd_kernel<<<b, 256, 0, st>>>(hidden, hp, mode, gptr, h, lin_w, val_w, (const float*)(sv + s.adv_off),
                                                  (const float*)(sv + s.v_off), dq, d_out_v, dh, (float*)(ws + w.dvr_off),
                                                  lpart, vpart, mask_dh);
    head_lin_grad_reduce_kernel<<<hidden + 1, 64, 0, st>>>(b, hp, hidden, lpart, d_lin_w, d_lin_b);
    if (has_value) head_lin_grad_reduce_kernel<<<hidden + 1, 64, 0, st>>>(b, hp, hidden, vpart, d_val_w, d_val_b);
    return check_launch();
}

int hexgnn_sage_scalar_forward(int n, int hidden, const int* rowptr, const int* col, const float* invdeg, const float* h,
                               const float* wl, const float* wr, const float* bias, float* out, float* dots,
                               hexgnn_stream_t stream_) {
    hipStream_t st = (hipStream_t)stream_;
    const int hp = padded_width(hidden);
    if (hp < 0) return HEXGNN_EUNSUPPORTED;
    if (n < 0 || !wl || !wr || !bias || (n > 0 && (!rowptr || !col || !invdeg || !h || !out || !dots))) return HEXGNN_EINVAL;
    if (n == 0) return HEXGNN_OK;
    sage_scalar_dots_kernel<<<(n + 63) / 64, 256, 0, st>>>(n, hidden, hp, h, wl, wr, dots);
    sage_scalar_gather_kernel<<<(n + 255) / 256, 256, 0, st>>>(n, rowptr, col, invdeg, dots, bias, out);
    return check_launch();
}

size_t hexgnn_sage_scalar_backward_workspace_bytes(int n, int hidden) {
    const int hp = padded_width(hidden);
    if (n < 0 || hp < 0) return 0;
    return 2 * align_up(sizeof(float) * (size_t)((n + 63) / 64 + 1) * (hp + 1), 256);
}

int hexgnn_sage_scalar_backward(int n, int hidden, const int* rowptr_t, const int* col_t, const float* invdeg,
                                const float* h, const float* wl, const float* wr, const float* dout, float* dh,
                                float* d_wl, float* d_wr, float* d_bias, void* workspace, size_t workspace_bytes,
                                hexgnn_stream_t stream_) {
    hipStream_t st = (hipStream_t)stream_;
    const int hp = padded_width(hidden);
    if (hp < 0) return HEXGNN_EUNSUPPORTED;
    if (n < 0 || !wl || !wr || !d_wl || !d_wr || !d_bias) return HEXGNN_EINVAL;
    if (n > 0 && (!rowptr_t || !col_t || !invdeg || !h || !dout || !dh)) return HEXGNN_EINVAL;
    if (!workspace || workspace_bytes < hexgnn_sage_scalar_backward_workspace_bytes(n, hidden)) return HEXGNN_EWORKSPACE;
    const int nblk = (n + 63) / 64;
    float* lpart = (float*)workspace;
    float* rpart = (float*)((char*)workspace + align_up(sizeof(float) * (size_t)(nblk + 1) * (hp + 1), 256));
    if (nblk > 0)
        sage_scalar_bwd_kernel<<<nblk, 256, 0, st>>>(n, hidden, hp, rowptr_t, col_t, invdeg, h, wl, wr, dout, dh, lpart, rpart);
    // (the right part's bias slot is zero: its sum lands in the scratch float behind the partials)
    head_lin_grad_reduce_kernel<<<hidden + 1, 64, 0, st>>>(nblk, hp, hidden, lpart, d_wl, d_bias);
    head_lin_grad_reduce_kernel<<<hidden + 1, 64, 0, st>>>(nblk, hp, hidden, rpart, d_wr, rpart + (size_t)nblk * (hp + 1));
    return check_launch();
}

int hexgnn_policy_log_softmax_forward(int n, int b, const int* gptr, const float* x, int x_stride, int swap_allowed,
                                      const float* pi_raw, const float* should_swap, float* out_pi, int64_t* out_gi,
                                      int64_t* out_ptr, hexgnn_stream_t stream_) {
    if (n < 0 || b < 0 || !out_ptr) return HEXGNN_EINVAL;
    if (b > 0 && (!gptr || !pi_raw || !out_pi || !out_gi)) return HEXGNN_EINVAL;
    if (swap_allowed && b > 0 && (!x || x_stride < 3 || !should_swap)) return HEXGNN_EINVAL;
    if (b == 0) { (void)hipMemsetAsync(out_ptr, 0, sizeof(int64_t), (hipStream_t)stream_); return check_launch(); }
    policy_lsm_fwd_kernel<<<b, 256, 0, (hipStream_t)stream_>>>(b, gptr, x, x_stride, swap_allowed, pi_raw, should_swap, out_pi,
                                                              out_gi, out_ptr);
    return check_launch();
}

int hexgnn_policy_log_softmax_backward(int n, int b, const int* gptr, const float* x, int x_stride, int swap_allowed,
                                       const int64_t* out_ptr, const float* out_pi, const float* d_out, float* d_pi_raw,
                                       float* d_should_swap, hexgnn_stream_t stream_) {
    if (n < 0 || b < 0) return HEXGNN_EINVAL;
    if (b > 0 && (!gptr || !out_ptr || !out_pi || !d_out || !d_pi_raw)) return HEXGNN_EINVAL;
    if (swap_allowed && b > 0 && (!x || x_stride < 3 || !d_should_swap)) return HEXGNN_EINVAL;
    if (b == 0) return HEXGNN_OK;
    policy_lsm_bwd_kernel<<<b, 256, 0, (hipStream_t)stream_>>>(b, gptr, x, x_stride, swap_allowed, out_ptr, out_pi, d_out,
                                                              d_pi_raw, d_should_swap);
    return check_launch();
}

}  // extern "C"
