// Batched weight-gradient GEMM of the hidden SAGE layers in exact fp32 (included by sage.hip and by the stand-alone
// timing harness tools/microbench/dw_coexec.hip).
//
// Reference: the autograd backward of SAGEConv's two linears (GN0/torch_script_models.py:52-73; call site GN0/models.py:276):
//   dW[o][i'] = sum_rows G[row][o] * [agg | x][row][i'],  db[o] = sum_rows G[row][o]
// grid (S row slices, hidden-input layers); the slabs are summed over the slices by the reduce kernel afterwards.
#pragma once
#include "hexgnn_internal.h"

namespace hexgnn {

struct DwArgs {
    const float* xin[kMaxLayers];
    const float* agg[kMaxLayers];
    const float* g[kMaxLayers];
    int n, rows_per_slice, S;
};

// Work split of one workgroup.  Wave w < NT owns output channels 16w..16w+15 (one 16-row tile of dW) against the first
// 2NT - E input-feature tiles of [agg | x]; for NT >= 4 one EXTRA wave owns the last E input-feature tiles for all NT output
// tiles (8 waves load the four SIMDs evenly instead of 7 waves loading them 2/2/2/1).
// VT > 0: the last VT of a regular wave's input-feature tiles are not run on the matrix pipe but as a register-tiled fp32
// FMA block on the VALU (4 x VT outputs per lane, operands from the same LDS chunk): the vector pipe is a second fp32 pipe
// of the same peak that otherwise idles in this kernel, and the k order of every sum is the MFMA's (ascending rows).
template <int NT, int VT = 0, int E = 2, int MV = 0> struct DwShape {
    static constexpr bool kBal = NT >= 4;
    static constexpr int kWaves = kBal ? NT + 1 : NT;
    static constexpr int kExt = kBal ? E : 0;                    // input-feature tiles of the extra wave
    static constexpr int kVal = VT;                              // ... of a regular wave's VALU block
    static constexpr int kRegB = 2 * NT - kExt - kVal;           // ... of a regular wave's MFMA part
    static_assert(kRegB >= 1 && (VT == 0 || VT == 2 || VT == 4), "tile split");
};

typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifndef HEXGNN_DW_RH
#define HEXGNN_DW_RH 16
#endif

// MV = 1 (NT >= 4, E = 2): the extra wave hands its last tile (output tile NT-1 x input tile 2NT-1) to wave 2, so that the SIMD
// pairs (w, w + 4) carry 24 / 24 / 25 / 25 MFMAs per k-step instead of 24 / 24 / 24 / 26.
template <int NT, int VT = 0, int E = 2, int MV = 0>
__global__ __launch_bounds__((64 * DwShape<NT, VT, E, MV>::kWaves)) void sage_dw_kernel(DwArgs a, float* __restrict__ part) {
    using SH = DwShape<NT, VT, E, MV>;
    constexpr bool kMove = MV == 1 && SH::kBal && E == 2 && VT == 0;
    constexpr int HP = 16 * NT;
    constexpr int RH = HEXGNN_DW_RH;                    // rows per chunk (a multiple of 4); two chunk buffers
    constexpr int AS = 2 * HP + 16;                     // == 16 (mod 32): conflict-free fragment reads
    constexpr int GS = (NT % 2 == 1) ? HP : HP + 16;
    constexpr int NTHR = 64 * SH::kWaves;
    constexpr int NB = SH::kRegB;
    constexpr int NE = SH::kExt;
    constexpr int VI = VT > 0 ? VT : 1;
    __shared__ __attribute__((aligned(16))) float As[2 * RH * AS];
    __shared__ __attribute__((aligned(16))) float Gs[2 * RH * GS];
    const int li = blockIdx.y, s = blockIdx.x;
    const float* __restrict__ xin = a.xin[li];
    const float* __restrict__ agg = a.agg[li];
    const float* __restrict__ gg = a.g[li];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const bool extra = SH::kBal && w == NT;     // wave-uniform
    const int r_beg = s * a.rows_per_slice;
    const int r_end = min(a.n, r_beg + a.rows_per_slice);

    constexpr int kAcc = (NB + (kMove ? 1 : 0)) > NE * NT ? NB + (kMove ? 1 : 0) : NE * NT;
    f32x4 acc[kAcc];         // regular wave: tile t < NB; extra wave: [NE * t + tb]
#pragma unroll
    for (int t = 0; t < kAcc; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    // VALU block of a regular wave: lane (oq = lane & 3, ig = lane >> 2) owns outputs o = 16w + 4oq + oo (oo < 4) x
    // i' = 16 NB + VT ig + j (j < VT)
    const int oq = lane & 3, ig = lane >> 2;
    float vacc[4][VI];
#pragma unroll
    for (int oo = 0; oo < 4; ++oo)
#pragma unroll
        for (int j = 0; j < VI; ++j) vacc[oo][j] = 0.f;

    // Double-buffered RH-row chunks (two workgroups per CU), ONE barrier per chunk: the global loads of chunk i+2 are
    // in flight and the LDS writes of chunk i+1 are issued ahead of chunk i's MFMAs and complete under them.  (Single-buffered
    // 32-row chunks, two barriers each, left the pipe idle while a workgroup staged: 216 -> 210 us at RH = 16.)
    // Staging loads go through raw buffer instructions: ONE 32-bit lane offset serves the three arrays (same [n][HP] layout),
    // rows at or beyond the slice end read as zeros by the hardware's range check (num_records = the slice end), so a chunk
    // costs one v_add per thread instead of three 64-bit multiply-adds, twelve zero moves and a compare -- and every VALU
    // instruction counts here: v_mfma_f32_16x16x4_f32 does not overlap VALU work on its SIMD (tools/microbench/
    // mfma_valu_overlap.hip: each VALU instruction adds ~3 cycles to the MFMA stream).
    constexpr int Q = NT * 4;
    constexpr int kPer = (RH * Q + NTHR - 1) / NTHR;
    const unsigned rec = (unsigned)r_end * (HP * 4);
    const __amdgpu_buffer_rsrc_t r_agg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(agg), 0, rec, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_xin = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xin), 0, rec, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gg), 0, rec, 0x00020000);
    unsigned goff[kPer], loffA[kPer], loffG[kPer];      // per thread: global byte offset inside a chunk, LDS float offsets
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        const int p = tid + NTHR * k;
        const int rr = p / Q, q = p % Q;
        goff[k] = p < RH * Q ? (unsigned)(rr * HP + 4 * q) * 4u : 0x80000000u;
        loffA[k] = rr * AS + 4 * q;
        loffG[k] = rr * GS + 4 * q;
    }
    f32x4 ra[kPer], rx[kPer], rg[kPer];
    auto issue = [&](int rc) {
        const unsigned base = (unsigned)rc * (HP * 4);
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const unsigned o = goff[k] + base;          // (a disabled thread stays out of range: 2^31 + base)
            ra[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_agg, o, 0, 0));
            rx[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_xin, o, 0, 0));
            rg[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_g, o, 0, 0));
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const int p = tid + NTHR * k;
            if (p < RH * Q) {
                *reinterpret_cast<f32x4*>(&As[buf * RH * AS + loffA[k]]) = ra[k];
                *reinterpret_cast<f32x4*>(&As[buf * RH * AS + loffA[k] + HP]) = rx[k];
                *reinterpret_cast<f32x4*>(&Gs[buf * RH * GS + loffG[k]]) = rg[k];
            }
        }
    };
    if (r_beg < r_end) {
        issue(r_beg);
        stage(0);
        if (r_beg + RH < r_end) issue(r_beg + RH);
    }
    __syncthreads();
    int buf = 0;
    for (int rc = r_beg; rc < r_end; rc += RH, buf ^= 1) {
        if (rc + RH < r_end) stage(buf ^ 1);                 // chunk i+1 -> the other buffer (its readers passed the last barrier)
        if (rc + 2 * RH < r_end) issue(rc + 2 * RH);
        const float* Ab = As + buf * RH * AS;
        const float* Gb = Gs + buf * RH * GS;
        if (!extra) {
#pragma unroll
            for (int ks = 0; ks < RH / 4; ++ks) {
                const float av = Gb[(4 * ks + kq) * GS + 16 * w + m];
                bsum += av;
#pragma unroll
                for (int t = 0; t < NB; ++t) {
                    const float bv = Ab[(4 * ks + kq) * AS + 16 * t + m];
                    acc[t] = mfma16x16x4(av, bv, acc[t]);
                }
                if constexpr (kMove) {
                    if (w == 2) {      // wave-uniform: the tile handed over by the extra wave
                        const float av2 = Gb[(4 * ks + kq) * GS + 16 * (NT - 1) + m];
                        const float bv2 = Ab[(4 * ks + kq) * AS + 16 * (2 * NT - 1) + m];
                        acc[NB] = mfma16x16x4(av2, bv2, acc[NB]);
                    }
                }
                if constexpr (VT > 0) {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const f32x4 gv = *reinterpret_cast<const f32x4*>(&Gb[(4 * ks + kk) * GS + 16 * w + 4 * oq]);
                        float xv[VI];
                        if constexpr (VT == 4) {
                            const f32x4 t4 = *reinterpret_cast<const f32x4*>(&Ab[(4 * ks + kk) * AS + 16 * NB + 4 * ig]);
                            xv[0] = t4[0]; xv[1] = t4[1]; xv[2] = t4[2]; xv[3] = t4[3];
                        } else {
                            const f32x2 t2 = *reinterpret_cast<const f32x2*>(&Ab[(4 * ks + kk) * AS + 16 * NB + 2 * ig]);
                            xv[0] = t2[0]; xv[1] = t2[1];
                        }
#pragma unroll
                        for (int oo = 0; oo < 4; ++oo)
#pragma unroll
                            for (int j = 0; j < VI; ++j) vacc[oo][j] = __builtin_fmaf(gv[oo], xv[j], vacc[oo][j]);
                    }
                }
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < RH / 4; ++ks) {
                float bv[NE > 0 ? NE : 1];
#pragma unroll
                for (int tb = 0; tb < NE; ++tb) bv[tb] = Ab[(4 * ks + kq) * AS + 16 * (2 * NT - NE + tb) + m];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float av = Gb[(4 * ks + kq) * GS + 16 * t + m];
#pragma unroll
                    for (int tb = 0; tb < NE; ++tb) {
                        if constexpr (kMove) { if (t == NT - 1 && tb == NE - 1) continue; }
                        acc[NE * t + tb] = mfma16x16x4(av, bv[tb], acc[NE * t + tb]);
                    }
                }
            }
        }
        __syncthreads();
    }
    // slab [HP][2HP] then bias [HP]
    float* slab = part + ((size_t)li * a.S + s) * ((size_t)HP * (2 * HP + 1));
    if (!extra) {
#pragma unroll
        for (int t = 0; t < NB; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) slab[(size_t)(16 * w + 4 * kq + q) * (2 * HP) + 16 * t + m] = acc[t][q];
        if constexpr (VT > 0) {
#pragma unroll
            for (int oo = 0; oo < 4; ++oo)
#pragma unroll
                for (int j = 0; j < VI; ++j)
                    slab[(size_t)(16 * w + 4 * oq + oo) * (2 * HP) + 16 * NB + VT * ig + j] = vacc[oo][j];
        }
        if constexpr (kMove) {
            if (w == 2) {
#pragma unroll
                for (int q = 0; q < 4; ++q) slab[(size_t)(16 * (NT - 1) + 4 * kq + q) * (2 * HP) + 16 * (2 * NT - 1) + m] = acc[NB][q];
            }
        }
        bsum += __shfl_xor(bsum, 16);
        bsum += __shfl_xor(bsum, 32);
        if (kq == 0) slab[(size_t)HP * 2 * HP + 16 * w + m] = bsum;
    } else {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int tb = 0; tb < NE; ++tb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if constexpr (kMove) { if (t == NT - 1 && tb == NE - 1) continue; }
                    slab[(size_t)(16 * t + 4 * kq + q) * (2 * HP) + 16 * (2 * NT - NE + tb) + m] = acc[NE * t + tb][q];
                }
    }
}

}  // namespace hexgnn
