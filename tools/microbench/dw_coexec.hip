// Micro-benchmark (not part of libhexgnn.so): the batched exact-fp32 weight-gradient GEMM (sage_dw_kernel, the shipped
// source: gnn_hex_amd/csrc/sage_dw_kernel.h) at the GNN-L Hex-11 B=256 shape (16 layers x 31 488 rows x 112 columns, 32 row
// slices) with part of every wave's output tiles moved from the matrix pipe to a register-tiled FMA block on the VALU.
//
//   variant <VT, E>: VT input-feature tiles of each regular wave on the VALU, E input-feature tiles on the extra wave
//   <0,2> is the round-3 kernel.  Every variant must produce the same slabs (compared word for word with <0,2>).
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I gnn_hex_amd/csrc -o tools/microbench/dw_coexec tools/microbench/dw_coexec.hip
//   ./tools/microbench/dw_coexec [reps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "sage_dw_kernel.h"

namespace hexgnn { thread_local int g_last_hip_error = 0; int g_prof_class = -1; void prof_begin(hipStream_t) {} void prof_end(hipStream_t) {} }
using namespace hexgnn;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#ifndef DW_S
#define DW_S 32
#endif
#ifndef DW_NT
#define DW_NT 7
#endif
#ifndef DW_N
#define DW_N 31488
#endif
#ifndef DW_LAYERS
#define DW_LAYERS 16
#endif
constexpr int NT = DW_NT, HP = 16 * NT, LAYERS = DW_LAYERS, N = DW_N, S = DW_S, RPS = ((N + S - 1) / S + 31) / 32 * 32;

template <int VT, int E, int MV = 0>
static float run(const DwArgs& a, float* part, int reps, hipStream_t st) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) sage_dw_kernel<NT, VT, E, MV><<<dim3(S, LAYERS), 64 * DwShape<NT, VT, E, MV>::kWaves, 0, st>>>(a, part);
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) sage_dw_kernel<NT, VT, E, MV><<<dim3(S, LAYERS), 64 * DwShape<NT, VT, E, MV>::kWaves, 0, st>>>(a, part);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / reps;
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 200;
    const size_t slab = (size_t)N * HP;
    float *acts, *agg, *G, *part, *part_ref;
    CK(hipMalloc(&acts, sizeof(float) * slab * (LAYERS + 1)));
    CK(hipMalloc(&agg, sizeof(float) * slab * LAYERS));
    CK(hipMalloc(&G, sizeof(float) * slab * LAYERS));
    const size_t pwords = (size_t)LAYERS * S * HP * (2 * HP + 1);
    CK(hipMalloc(&part, sizeof(float) * pwords));
    CK(hipMalloc(&part_ref, sizeof(float) * pwords));
    {
        std::vector<float> h(slab * (LAYERS + 1));
        unsigned s = 12345u;
        auto fill = [&](float* d, size_t n, float scale) {
            for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = scale * ((int)(s >> 8) % 2001 - 1000) / 1000.f; }
            // ReLU-like sparsity (about half the activations / masked gradients are exactly zero in the real step)
            for (size_t i = 0; i < n; ++i) if (h[i] < 0.f && scale > 0.5f) h[i] = 0.f;
            CK(hipMemcpy(d, h.data(), sizeof(float) * n, hipMemcpyHostToDevice));
        };
        fill(acts, slab * (LAYERS + 1), 1.f);
        fill(agg, slab * LAYERS, 1.f);
        fill(G, slab * LAYERS, 0.01f);
    }
    DwArgs a;
    memset(&a, 0, sizeof(a));
    for (int l = 0; l < LAYERS; ++l) { a.xin[l] = acts + slab * l; a.agg[l] = agg + slab * l; a.g[l] = G + slab * l; }
    a.n = N; a.rows_per_slice = RPS; a.S = S;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    std::vector<float> ref(pwords), got(pwords);
    auto check = [&](const char* name) {
        CK(hipMemcpy(got.data(), part, sizeof(float) * pwords, hipMemcpyDeviceToHost));
        size_t bad = 0; double worst = 0.;
        for (size_t i = 0; i < pwords; ++i) {
            if (memcmp(&got[i], &ref[i], 4) != 0) { ++bad; const double d = fabs((double)got[i] - ref[i]); if (d > worst) worst = d; }
        }
        printf("  %s vs <0,2>: %zu of %zu words differ (max abs %.3g)\n", name, bad, pwords, worst);
    };
    const double flop = 2.0 * LAYERS * (double)N * HP * (2 * HP + 1);
    // warm the clocks, then alternate the variants a few times (boxes drift)
#if DW_NT < 4
    for (int round = 0; round < 3; ++round) {
        const float t = run<0, 2>(a, part_ref, reps, st);
        printf("round %d  NT %d N %d layers %d S %d RH %d: %7.1f us  %6.1f TFLOP/s\n", round, NT, N, LAYERS, S, HEXGNN_DW_RH, t, flop / t * 1e-6);
    }
    return 0;
#endif
    for (int round = 0; round < 3; ++round) {
        float t;
        t = run<0, 2>(a, part_ref, reps, st); printf("round %d  <VT 0, E 2> %7.1f us  %6.1f TFLOP/s\n", round, t, flop / t * 1e-6);
        if (round == 0) CK(hipMemcpy(ref.data(), part_ref, sizeof(float) * pwords, hipMemcpyDeviceToHost));
        CK(hipMemset(part, 0xff, sizeof(float) * pwords));
        t = run<2, 2>(a, part, reps, st); printf("round %d  <VT 2, E 2> %7.1f us  %6.1f TFLOP/s\n", round, t, flop / t * 1e-6);
        if (round == 0) check("<2,2>");
        CK(hipMemset(part, 0xff, sizeof(float) * pwords));
        t = run<2, 1>(a, part, reps, st); printf("round %d  <VT 2, E 1> %7.1f us  %6.1f TFLOP/s\n", round, t, flop / t * 1e-6);
        if (round == 0) check("<2,1>");
        CK(hipMemset(part, 0xff, sizeof(float) * pwords));
        t = run<4, 2>(a, part, reps, st); printf("round %d  <VT 4, E 2> %7.1f us  %6.1f TFLOP/s\n", round, t, flop / t * 1e-6);
        if (round == 0) check("<4,2>");
        CK(hipMemset(part, 0xff, sizeof(float) * pwords));
        t = run<4, 1>(a, part, reps, st); printf("round %d  <VT 4, E 1> %7.1f us  %6.1f TFLOP/s\n", round, t, flop / t * 1e-6);
        if (round == 0) check("<4,1>");
        t = run<0, 1>(a, part, reps, st); printf("round %d  <VT 0, E 1> %7.1f us  %6.1f TFLOP/s\n", round, t, flop / t * 1e-6);
        if (round == 0) check("<0,1>");
        CK(hipMemset(part, 0xff, sizeof(float) * pwords));
        t = run<0, 2, 1>(a, part, reps, st); printf("round %d  <VT 0, E 2, MV 1> %7.1f us  %6.1f TFLOP/s\n", round, t, flop / t * 1e-6);
        if (round == 0) check("<0,2,1>");
        t = run<0, 2>(a, part_ref, reps, st); printf("round %d  <VT 0, E 2> again %7.1f us  %6.1f TFLOP/s\n", round, t, flop / t * 1e-6);
    }
    return 0;
}
