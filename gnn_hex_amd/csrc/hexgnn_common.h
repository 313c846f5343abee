// Shared device/host helpers for libhexgnn.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/hexgnn.h"

namespace hexgnn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;        // CDNA wavefront
constexpr int kMaxNT = 8;        // hidden <= 128
constexpr int kSmallCin = 8;     // raw-feature first layer: c_in <= 8, stored padded to 8

extern thread_local int g_last_hip_error;

inline int check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last_hip_error = (int)e; return HEXGNN_EHIP; }
    return HEXGNN_OK;
}

// RAII event pair around one launch of a profiled kernel class (profile.hip)
extern int g_prof_class;
void prof_begin(hipStream_t st);
void prof_end(hipStream_t st);
struct KernelTimer {
    hipStream_t st; bool on;
    KernelTimer(int cls, hipStream_t s) : st(s), on(cls == g_prof_class) { if (on) prof_begin(st); }
    ~KernelTimer() { if (on) prof_end(st); }
};

inline int padded_width(int hidden) {
    if (hidden <= 0 || hidden > 16 * kMaxNT) return -1;
    return 16 * ((hidden + 15) / 16);
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// v_mfma_f32_16x16x4_f32: D[16x16] += A[16x4] * B[4x16], exact fp32 (k-ordered fmaf chain).
// lane l: a = A[l&15][l>>4], b = B[l>>4][l&15]; acc[r] = D[4*(l>>4)+r][l&15].
__device__ __forceinline__ f32x4 mfma16x16x4(float a, float b, f32x4 acc) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
}

// Power-of-two scale for a block of values whose largest magnitude is m >= 0: m*s lies in [2^14, 2^15), so the fp16 hi
// part never overflows and the lo parts of all but negligible elements stay normal (split-precision "f16x3" math).
__device__ __forceinline__ void pow2_scale(const float m, float& s, float& inv) {
    const unsigned e = __builtin_bit_cast(unsigned, m) >> 23;    // biased exponent
    const bool ok = e >= 64u && e <= 190u;                       // 2^-63 <= m < 2^64; anything else stays unscaled
    s = ok ? __builtin_bit_cast(float, (268u - e) << 23) : 1.f;
    inv = ok ? __builtin_bit_cast(float, (e - 14u) << 23) : 1.f;
}

// d/dx tanh(x) = sech^2(x) = 4e/(1+e)^2 with e = exp(-2|x|): no cancellation, so saturated advantages / values keep
// a few-ulp derivative (1 - tanh(x)^2 loses all relative accuracy as |tanh| -> 1).
__device__ __forceinline__ float sech2f(float x) {
    const float e = expf(-2.f * fabsf(x));
    const float d = 1.f + e;
    return 4.f * e / (d * d);
}

}  // namespace hexgnn
