// Micro-benchmark (not part of libhexgnn.so): one fused GNN-L layer loop in two workgroup shapes, to cost the
// one-wave-per-SIMD layout DESIGN.md section 7.3 proposes for graphs above 128 rows before anyone builds it.
//
//   shape A  W = 8 waves x M = 1 row tile   (today's qnet_fwd_kernel: two waves share a SIMD's MFMA pipe)
//   shape B  W = 4 waves x M = 2 row tiles  (one wave per SIMD, 512 registers; weight fragments read once for both tiles)
//   shape C  W = 4 waves x M = 3 row tiles  (192 rows; register / scheduling behaviour only: the row buffer is folded onto
//            128 rows because 192 rows + two 50-KB weight halves exceed the LDS -- the real kernel needs K-quarter buffers)
//
// Per layer and wave the loop does what the forward kernel's layer does, with synthetic operands: self-half contraction
// (v_mfma_f32_16x16x4_f32, weights from LDS, rows from registers) carrying the LDS gather of six neighbour rows per row, the
// previous rows' global stores and the LDS-DMA of the other weight half as ONE micro-op per MFMA slot; barrier; aggregate-half
// contraction carrying the aggregate's stores and the next half's DMA; bias + ReLU epilogue into registers and LDS; barrier.
// Prints us per launch / per layer and the fraction of the pure MFMA time (W/4 x M x 392 MFMAs x 32 cycles per SIMD and layer
// at the measured shader clock is not known here: 2.4 GHz assumed, compare the shapes with each other).
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o layer_shapes tools/microbench/layer_shapes.hip && ./layer_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <utility>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int NT = 7, HP = 16 * NT, XS = HP + 4, kHalf = NT * NT * 64;   // float4 per weight half
constexpr int KN = 6;                                                    // neighbours gathered per row
constexpr unsigned kOob = 0x80000000u;

template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) { f(std::integral_constant<int, B>{}); static_for<B + 1, E>(f); }
}
__device__ __forceinline__ f32x4 mfma(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void wait_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
typedef unsigned u32x4b __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t slab_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ void buf_store(const f32x4 v, __amdgpu_buffer_rsrc_t r, unsigned off) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4b, v), r, off, 0, 0);
}
__device__ __forceinline__ void dma_piece(const void* gsrc_piece, unsigned lane_off, unsigned lds_dst) {
    unsigned keep;
    lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);
    const unsigned long long sb = (unsigned long long)gsrc_piece;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)sb), hi = __builtin_amdgcn_readfirstlane((unsigned)(sb >> 32));
    const unsigned long long sbase = ((unsigned long long)hi << 32) | lo;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_off), "s"(lds_dst), "s"(sbase) : "memory");
}

// FINE: one filler micro-op after every MFMA (the one-wave shape needs it: nobody else feeds the pipe while a wave runs a
// block of fillers); otherwise the fillers of a (chunk, k-step) group run as one block behind the group's MFMAs (today's kernels)
template <int W, int M, bool FINE, int ROWS>
__global__ __launch_bounds__(64 * W) void layer_kernel(const f32x4* __restrict__ wsrc, float* __restrict__ out, int layers, int n_rows) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    f32x4* wbuf = reinterpret_cast<f32x4*>(lds);                       // [2][kHalf]
    float* xbuf = reinterpret_cast<float*>(lds + 2 * kHalf * 16);      // [ROWS + 1][XS]
    float* s_bias = xbuf + (ROWS + 1) * XS;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    constexpr int kSlots = 4 * NT * M * NT;          // MFMAs per wave and phase
    constexpr int kOps = 2 * NT * KN * M;            // gather micro-ops per wave (7 reads + 7 adds per neighbour)
    constexpr int kDma = (NT * NT + W - 1) / W;      // LDS-DMA pieces per wave and half
    static_assert(kOps + kDma + NT * M <= kSlots, "fillers must fit the slots");

    for (int i = tid; i < 2 * kHalf; i += 64 * W) wbuf[i] = wsrc[i];
    if (tid < HP) s_bias[tid] = 0.001f * (float)tid;
    if (tid < XS) xbuf[ROWS * XS + tid] = 0.f;
    f32x4 xs[M][NT];
    unsigned noff[M][KN];
    int lrow[M];
#pragma unroll
    for (int tl = 0; tl < M; ++tl) {
        lrow[tl] = (wave * M + tl) * 16 + r;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            xs[tl][t] = f32x4{0.01f * (float)(lrow[tl] % 13), 0.02f, 0.001f * (float)t, 0.003f * (float)g};
            reinterpret_cast<f32x4*>(xbuf + (lrow[tl] % ROWS) * XS)[4 * t + g] = xs[tl][t];
        }
#pragma unroll
        for (int k = 0; k < KN; ++k) noff[tl][k] = (unsigned)(((lrow[tl] * 7 + k * 13 + 5) % ROWS) * (XS * 4) + 16 * g);
    }
    __syncthreads();
    const unsigned lds_w = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    const unsigned lane16 = 16 * lane;
    const char* xb = reinterpret_cast<const char*>(xbuf);
    const size_t slab = (size_t)n_rows * HP;
    const int grow0 = blockIdx.x * (16 * W * M);

    for (int l = 1; l < layers; ++l) {
        f32x4 acc[M][NT], ag[M][NT], land[NT];
#pragma unroll
        for (int tl = 0; tl < M; ++tl)
#pragma unroll
            for (int t = 0; t < NT; ++t) { acc[tl][t] = f32x4{0.f, 0.f, 0.f, 0.f}; ag[tl][t] = acc[tl][t]; }
        const __amdgpu_buffer_rsrc_t yprev = slab_rsrc(out + slab * (size_t)((l - 1) & 1));
        const __amdgpu_buffer_rsrc_t aout = slab_rsrc(out + slab * (size_t)(2 + (l & 1)));

        // one filler micro-op: index o of the phase's list
        auto opS = [&](auto oo) {
            constexpr int o = decltype(oo)::value;
            if constexpr (o < kOps) {
                constexpr int tl = o / (2 * NT * KN), k = (o % (2 * NT * KN)) / (2 * NT), sub = o % (2 * NT);
                if constexpr (sub < NT) land[sub] = *reinterpret_cast<const f32x4*>(xb + noff[tl][k] + 64 * sub);
                else ag[tl][sub - NT] += land[sub - NT];
            } else if constexpr (o < kOps + kDma) {
                const int p = wave + W * (o - kOps);
                if (p < NT * NT) dma_piece(wsrc + p * 64, lane16, lds_w + p * 1024);
            } else if constexpr (o < kOps + kDma + NT * M) {
                constexpr int q = o - kOps - kDma, tl = q / NT, t = q % NT;
                buf_store(xs[tl][t], yprev, (unsigned)(grow0 + lrow[tl]) * (HP * 4) + 16 * g + 64 * t);
            }
        };
        auto opA = [&](auto oo) {
            constexpr int o = decltype(oo)::value;
            if constexpr (o < kDma) {
                const int p = wave + W * o;
                if (p < NT * NT) dma_piece(wsrc + kHalf + p * 64, lane16, lds_w + (kHalf + p * 64) * 16);
            } else if constexpr (o < kDma + NT * M) {
                constexpr int q = o - kDma, tl = q / NT, t = q % NT;
                buf_store(ag[tl][t], aout, (unsigned)(grow0 + lrow[tl]) * (HP * 4) + 16 * g + 64 * t);
            }
        };
        // K-half contraction; `ops` micro-ops spread evenly over the kSlots MFMA slots (FINE) or over the 4 NT groups
        auto contract = [&](const f32x4* whalf, const f32x4 (&x)[M][NT], auto nops, auto&& op) {
            constexpr int kN = decltype(nops)::value;
            f32x4 w[2][NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) w[0][t] = whalf[t * 64 + lane];
            static_for<0, NT>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                static_for<0, 4>([&](auto jj) {
                    constexpr int j = decltype(jj)::value;
                    if constexpr (W == 4 && j == 1 && c + 1 < NT) {     // one wave per SIMD: second fragment set, requested early
#pragma unroll
                        for (int t = 0; t < NT; ++t) w[(c + 1) & 1][t] = whalf[((c + 1) * NT + t) * 64 + lane];
                    }
                    static_for<0, M * NT>([&](auto ss) {
                        constexpr int s = decltype(ss)::value, tl = s / NT, t = s % NT;
                        constexpr int slot = (4 * c + j) * M * NT + s;
                        acc[tl][t] = mfma(w[W == 4 ? (c & 1) : 0][t][j], x[tl][c][j], acc[tl][t]);
                        if constexpr (FINE) {
                            __builtin_amdgcn_sched_barrier(0);
                            static_for<(slot * kN + kSlots - 1) / kSlots, ((slot + 1) * kN + kSlots - 1) / kSlots>(op);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    });
                    if constexpr (!FINE) {
                        constexpr int grp = 4 * c + j, kG = 4 * NT;
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (W != 4 && j == 3 && c + 1 < NT) {
#pragma unroll
                            for (int t = 0; t < NT; ++t) w[0][t] = whalf[((c + 1) * NT + t) * 64 + lane];
                        }
                        static_for<(grp * kN + kG - 1) / kG, ((grp + 1) * kN + kG - 1) / kG>(op);
                        __builtin_amdgcn_sched_barrier(0);
                    } else if constexpr (W != 4 && j == 3 && c + 1 < NT) {
#pragma unroll
                        for (int t = 0; t < NT; ++t) w[0][t] = whalf[((c + 1) * NT + t) * 64 + lane];
                    }
                });
            });
        };
        contract(wbuf + kHalf, xs, std::integral_constant<int, kOps + kDma + NT * M>{}, opS);
#pragma unroll
        for (int tl = 0; tl < M; ++tl)
#pragma unroll
            for (int t = 0; t < NT; ++t) ag[tl][t] *= 0.1666f;
        wait_vmem();
        lds_barrier();
        contract(wbuf, ag, std::integral_constant<int, kDma + NT * M>{}, opA);
#pragma unroll
        for (int tl = 0; tl < M; ++tl) {
            f32x4* xr = reinterpret_cast<f32x4*>(xbuf + (lrow[tl] % ROWS) * XS) + g;
            const f32x4* bl = reinterpret_cast<const f32x4*>(s_bias) + g;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                f32x4 v = acc[tl][t] + bl[4 * t];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
                xs[tl][t] = v;
                xr[4 * t] = v;
            }
        }
        wait_vmem();
        lds_barrier();
    }
#pragma unroll
    for (int tl = 0; tl < M; ++tl)
#pragma unroll
        for (int t = 0; t < NT; ++t)
            reinterpret_cast<f32x4*>(out + 4 * slab + (size_t)(grow0 + lrow[tl]) * HP)[4 * t + g] = xs[tl][t];
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int W, int M, bool FINE, int ROWS>
static void run(const char* name, const f32x4* wsrc, float* out, int layers, int graphs, int n_rows) {
    const size_t lds = (size_t)2 * kHalf * 16 + (size_t)(ROWS + 1) * XS * 4 + HP * 4;
    auto kern = layer_kernel<W, M, FINE, ROWS>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) kern<<<graphs, 64 * W, lds>>>(wsrc, out, layers, n_rows);
    CK(hipDeviceSynchronize());
    const int reps = 50;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) kern<<<graphs, 64 * W, lds>>>(wsrc, out, layers, n_rows);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / reps, per_layer = us / (layers - 1);
    const double mfma_us = (double)(W / 4) * M * 2 * 4 * NT * NT * 32 / 2.4e3;    // per layer and SIMD at 2.4 GHz
    printf("%-34s rows/wg %3d  lds %6zu  %8.1f us/launch  %6.2f us/layer  %5.2f us/layer per 16-row tile-pair  mfma-only %5.2f us (%.0f %%)\n",
           name, 16 * W * M, lds, us, per_layer, per_layer * 2.0 / ((W / 4) * M), mfma_us, 100.0 * mfma_us / per_layer);
}

int main() {
    const int layers = 15, graphs = 256;
    const int n_rows = graphs * 192;
    f32x4* wsrc; float* out;
    CK(hipMalloc(&wsrc, (size_t)2 * kHalf * 16));
    CK(hipMalloc(&out, (size_t)5 * n_rows * HP * 4));
    std::vector<float> hw((size_t)2 * kHalf * 4);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = 0.01f * (float)((i * 2654435761u >> 20) % 17) - 0.08f;
    CK(hipMemcpy(wsrc, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    run<8, 1, false, 128>("A  8 waves x 1 tile, group fillers", wsrc, out, layers, graphs, n_rows);
    run<8, 1, true, 128>("A' 8 waves x 1 tile, fine fillers", wsrc, out, layers, graphs, n_rows);
    run<4, 2, true, 128>("B  4 waves x 2 tiles, fine fillers", wsrc, out, layers, graphs, n_rows);
    run<4, 2, false, 128>("B' 4 waves x 2 tiles, group fillers", wsrc, out, layers, graphs, n_rows);
    run<4, 3, true, 128>("C  4 waves x 3 tiles (rows folded)", wsrc, out, layers, graphs, n_rows);
    return 0;
}
