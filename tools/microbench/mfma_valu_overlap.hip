// Micro-benchmark: does VALU fp32 work overlap v_mfma_f32_16x16x4_f32 on one SIMD?  (Round-4 question behind "VALU co-execution"
// in sage_dw_kernel: the fp32 MFMA runs at exactly the fp32 VECTOR rate -- is it a second pipe or the same multipliers?)
//
// One workgroup per CU, W waves per SIMD; every wave runs ITER iterations of 8 independent-accumulator MFMAs, with M VALU
// instructions (register operands only, independent chains) placed behind each MFMA.  Prints cycles per MFMA slot per SIMD.
//   kind 0: v_pk_fma_f32   kind 1: v_fma_f32   kind 2: v_pk_add_f32   kind 3: v_add_f32   kind 4: v_mov_b32 (issue only)
//   kind 5: ds_read_b128 (results never used)   kind 6: s_add_u32   kind 7: v_max_f32   kind 8: v_cndmask_b32
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/microbench/mfma_valu_overlap tools/microbench/mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int M, int KIND, bool MFMA>
__global__ __launch_bounds__(1024) void k(float* out, int iters, unsigned long long* ticks) {
    f32x4 acc[8];
    f32x2 v[8];
    const float a = (float)threadIdx.x * 1e-3f, b = 1.0001f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { acc[i] = f32x4{a, a, a, a}; v[i] = f32x2{a + i, a - i}; }
    const f32x2 c = f32x2{1.00001f, 0.99999f}, d = f32x2{1e-6f, -1e-6f};
    __shared__ float lbuf[4096];
    lbuf[threadIdx.x] = a; lbuf[threadIdx.x + 1024] = a;
    __syncthreads();
    const unsigned ldsaddr = (threadIdx.x & 63) * 16;
    unsigned sreg = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if constexpr (MFMA) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < M; ++j) {
                f32x2& r = v[(i * M + j) & 7];
                if constexpr (KIND == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(c), "v"(d));
                else if constexpr (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[0]) : "v"(c[0]), "v"(d[0]));
                else if constexpr (KIND == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(r) : "v"(d));
                else if constexpr (KIND == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[0]) : "v"(d[0]));
                else if constexpr (KIND == 4) asm volatile("v_mov_b32 %0, %1" : "+v"(r[0]) : "v"(d[0]));
                else if constexpr (KIND == 5) { f32x4 t; asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(ldsaddr)); }   // never waited for
                else if constexpr (KIND == 6) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sreg));
                else if constexpr (KIND == 7) asm volatile("v_max_f32 %0, %0, %1" : "+v"(r[0]) : "v"(d[0]));
                else if constexpr (KIND == 8) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[0]) : "v"(d[0]));
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3] + v[i][0] + v[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)sreg;
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

template <int M, int KIND, bool MFMA>
static void run(int waves_per_simd, float* out, unsigned long long* ticks, const char* name) {
    const int iters = 2000, threads = 256 * waves_per_simd;
    k<M, KIND, MFMA><<<256, threads>>>(out, iters, ticks);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    k<M, KIND, MFMA><<<256, threads>>>(out, iters, ticks);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long t = 0;
    CK(hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost));
    const double slots = (double)iters * 8 * waves_per_simd;      // MFMA slots per SIMD
    printf("%-12s M=%d mfma=%d waves/SIMD=%d: %8.1f us, %6.1f ticks per MFMA slot per SIMD (wave 0: %llu ticks)\n", name, M, (int)MFMA,
           waves_per_simd, ms * 1e3, (double)t / slots, t);
}

int main() {
    float* out; unsigned long long* ticks;
    CK(hipMalloc(&out, sizeof(float) * 256 * 1024));
    CK(hipMalloc(&ticks, 64));
    for (int w : {1, 2, 4}) {
        run<0, 0, true>(w, out, ticks, "mfma only");
        run<1, 0, true>(w, out, ticks, "pk_fma"); run<2, 0, true>(w, out, ticks, "pk_fma"); run<4, 0, true>(w, out, ticks, "pk_fma");
        run<6, 0, true>(w, out, ticks, "pk_fma");
        run<2, 1, true>(w, out, ticks, "fma"); run<4, 1, true>(w, out, ticks, "fma"); run<6, 1, true>(w, out, ticks, "fma");
        run<2, 2, true>(w, out, ticks, "pk_add"); run<4, 2, true>(w, out, ticks, "pk_add");
        run<2, 3, true>(w, out, ticks, "add"); run<4, 3, true>(w, out, ticks, "add");
        run<4, 4, true>(w, out, ticks, "mov"); run<6, 4, true>(w, out, ticks, "mov");
        run<1, 5, true>(w, out, ticks, "ds_read_b128"); run<2, 5, true>(w, out, ticks, "ds_read_b128"); run<4, 5, true>(w, out, ticks, "ds_read_b128");
        run<2, 6, true>(w, out, ticks, "s_add"); run<6, 6, true>(w, out, ticks, "s_add");
        run<4, 7, true>(w, out, ticks, "v_max"); run<4, 8, true>(w, out, ticks, "v_cndmask");
        run<4, 0, false>(w, out, ticks, "pk_fma alone"); run<4, 1, false>(w, out, ticks, "fma alone");
    }
    return 0;
}
