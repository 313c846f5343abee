// Weight packing of a SAGE stack (one launch per forward call: sage_pack_kernel in sage.hip, or the extra workgroups of
// csr_grouped_pack_kernel in csr.hip -- the CSR build and the pack do not depend on each other).
#pragma once
#include "hexgnn_internal.h"

namespace hexgnn {

struct LayerPtrs {
    const float* wl[kMaxLayers];
    const float* bl[kMaxLayers];
    const float* wr[kMaxLayers];
};

// ---- weight packing (one launch per stack call; grid.y = layer) ------------------------------------
struct PackArgs {
    LayerPtrs p;
    size_t fwd_off[kMaxLayers], bwd_off[kMaxLayers], bias_off[kMaxLayers];
    size_t flag_off;
    int hp, nt, L, c_in, hidden, small_first;
};

__device__ __forceinline__ void sage_pack_body(const PackArgs& a, char* __restrict__ wpack, int bx, int l, int nbx) {
    const int hp = a.hp, nt = a.nt, H = a.hidden;
    const float* wl = a.p.wl[l];
    const float* wr = a.p.wr[l];
    const float* bl = a.p.bl[l];
    float* bias = (float*)(wpack + a.bias_off[l]);
    const int tid = bx * 256 + (int)threadIdx.x;            // (256-thread workgroups, nbx of them per layer)
    if (tid < hp) bias[tid] = tid < H ? bl[tid] : 0.f;
    // progress counters of the one-launch stack kernels (forward + backward) start every forward call at zero
    if (l == 0) {
        unsigned* flags = reinterpret_cast<unsigned*>(wpack + a.flag_off);
        for (int i = tid; i < 2 * kStackFlagWords; i += nbx * 256) flags[i] = 0u;
    }
    if (l == 0 && a.small_first) {
        float* w0 = (float*)(wpack + a.fwd_off[l]);
        const int tot = hp * kSmallCin;
        if (tid < tot) {
            const int o = tid / kSmallCin, q = tid % kSmallCin;
            const bool ok = o < H && q < a.c_in;
            w0[tid] = ok ? wl[o * a.c_in + q] : 0.f;
            w0[tot + tid] = ok ? wr[o * a.c_in + q] : 0.f;
        }
        return;
    }
    const int in = H;  // hidden -> hidden
    const int tot = 2 * nt * nt * 256;
    if (tid >= tot) return;
    float* pf = (float*)(wpack + a.fwd_off[l]);
    float* pb = (float*)(wpack + a.bwd_off[l]);
    {   // forward pack  P[c][t][lane][j], c < 2NT (k chunk of [agg|x]), t < NT (output tile)
        const int c = tid / (nt * 256), rem = tid % (nt * 256);
        const int t = rem / 256, lj = rem % 256, lane = lj >> 2, j = lj & 3;
        const int g = lane >> 4, cx = lane & 15;
        const int k = 16 * (c % nt) + 4 * g + j, o = 16 * t + cx;
        const float* w = c < nt ? wl : wr;
        pf[tid] = (k < in && o < H) ? w[o * in + k] : 0.f;
    }
    {   // backward pack PB[h][c][t][lane][j]: h = 0 the W_l part (dAgg), 1 the W_r part (dXs), each a contiguous half;
        // c < NT (k chunk over outputs o), t < NT (tile of input features)
        const int h = tid / (nt * nt * 256), rem0 = tid % (nt * nt * 256);
        const int c = rem0 / (nt * 256), rem = rem0 % (nt * 256);
        const int t = rem / 256, lj = rem % 256, lane = lj >> 2, j = lj & 3;
        const int g = lane >> 4, cx = lane & 15;
        const int o = 16 * c + 4 * g + j, i = 16 * t + cx;
        const float* w = h == 0 ? wl : wr;
        pb[tid] = (o < H && i < in) ? w[o * in + i] : 0.f;
    }
}



// host side: fill PackArgs from a plan (sage.hip)
int fill_pack_args(const StackPlan& p, int c_in, int hidden, const float* const* wl, const float* const* bl,
                   const float* const* wr, PackArgs* pa);

}  // namespace hexgnn
