// Cross-translation-unit internals of libhexgnn.so (plans + launch helpers shared by the layer-major
// path in sage.hip / head.hip and the fused per-graph path in qnet_fused.hip).
#pragma once
#include "hexgnn_common.h"

namespace hexgnn {

constexpr int kMaxLayers = 64;
constexpr int kStackFlagWords = 512;   // row blocks a one-launch stack kernel can track (>= CUs of the device)
constexpr int kDwMaxSlices = 64;   // row slices per layer of the weight-gradient GEMM (workspace is sized for this many)
// slabs a plan with `layers` layers must hold: kDwMaxSlices per layer, and one- / two-layer launches use up to twice that
inline size_t dw_slab_count(int layers) {
    const size_t a = (size_t)layers * kDwMaxSlices, b = 4 * (size_t)kDwMaxSlices;
    return a > b ? a : b;
}

struct StackPlan {
    int hp, nt, L, c_in;
    bool small_first;
    size_t fwd_off[kMaxLayers];   // byte offsets into wpack
    size_t bwd_off[kMaxLayers];
    size_t bias_off[kMaxLayers];
    size_t flag_off;              // 2 x kStackFlagWords unsigned (forward, backward): per-block progress counters of the
    size_t pack_bytes;            // one-launch stack kernels
    size_t agg_off[kMaxLayers];   // byte offsets into saved
    size_t saved_bytes;
};
int make_plan(int n, int c_in, int hidden, int L, StackPlan* p);

struct BwdPlan {
    size_t g_off, part_off, part0_off, total;
    int S, rps;
    int S0, rps0;   // raw first layer: many small slices (VALU kernel, one pass over G)
};
void make_bwd_plan(int n, const StackPlan& p, BwdPlan* b);
int dw_slices_for(int n, int hidden_layers, int math, int stack_hidden_layers);   // slices the weight-gradient launch actually uses
// (<= BwdPlan::S; hidden_layers = layers of this launch, stack_hidden_layers = of the whole stack: only a stack of <= 2 gets the
// doubled slice count, so that a stage of a longer stack stays inside its kDwMaxSlices-per-layer slab region)

// pack all layers of a stack (forward + backward fragment order, padded bias) -- one launch
int launch_pack(const StackPlan& p, int c_in, int hidden, const float* const* wl, const float* const* bl,
                const float* const* wr, void* wpack, hipStream_t st, int math = 0, unsigned* zero_maxima = nullptr);
// batched weight-gradient GEMM + reductions for all layers of a stack, given G (per-layer masked output gradients)
int launch_weight_grads(int n, int c_in, int hidden, const StackPlan& p, const BwdPlan& b, const float* x,
                        int x_stride, const float* acts, const char* saved, const float* G, float* const* d_wl,
                        float* const* d_bl, float* const* d_wr, float* part, float* part0, hipStream_t st,
                        int math = 0, const unsigned* xmax = nullptr, const unsigned* gmax = nullptr,
                        bool hidden_only_no_reduce = false, int layer_lo = -1, int layer_hi = -1);
// layer_lo / layer_hi (>= 0): only the hidden-input layers [layer_lo, layer_hi) -- staged backward; `part` then is the
// stage's own slab region and the slice count follows the stage's layer count (dw_slices_for(n, layer_hi - layer_lo, math))
// hidden_only_no_reduce: leave the slice slabs in `part` and skip the raw first layer (the fused path reduces everything in
// one launch and gets the first layer's per-graph partials from its backward kernel).
// math 1: f16x3 split with per-layer power-of-two scales from xmax[l] = max |[agg_l | x_l]|, gmax[l] = max |G_l| (bit patterns)

// ---- hidden 129..256 (wide.hip): plain kernels behind the same entry points ------------------------------------------------
constexpr int kWideMaxHidden = 256;
constexpr int kWideSlicesMax = 64;           // row slices of a wide weight-gradient launch (workspace sized for this many)
int padded_width_wide(int hidden);           // 16-multiple up to 256, -1 beyond
struct WidePlan {
    int hp, L;
    bool small_first;
    size_t w_off[kMaxLayers], bias_off[kMaxLayers], pack_bytes;
    size_t agg_off[kMaxLayers], saved_bytes;
    size_t g_off, tmp_off, part_off, bwd_bytes;
};
int wide_make_plan(int n, int c_in, int hidden, int L, WidePlan* p);
int wide_stack_forward(int n, int c_in, int hidden, int L, const int* rowptr, const int* col, const float* invdeg,
                       const float* x, int x_stride, const float* const* wl, const float* const* bl,
                       const float* const* wr, void* wpack, float* acts, void* saved, int flags, hipStream_t st);
int wide_stack_backward(int n, int c_in, int hidden, int L, const int* rowptr_t, const int* col_t, const float* invdeg,
                        const float* x, int x_stride, const float* acts, const void* saved, const void* wpack,
                        const float* dy, float* dx, float* const* d_wl, float* const* d_bl, float* const* d_wr,
                        void* workspace, size_t workspace_bytes, int flags, int tap_layer, float* tap_out, hipStream_t st);
int wide_head_forward(int n, int b, int hidden, int mode, const int* gptr, const float* h, const float* lin_w,
                      const float* lin_b, const float* v0_w, const float* v0_b, const float* v1_w, const float* v1_b,
                      float* q, float* out_v, void* saved, hipStream_t st);
int wide_head_backward(int n, int b, int hidden, int mode, const int* gptr, const float* h, const float* lin_w,
                       const float* v0_w, const float* v1_w, const void* saved, const float* dq, const float* d_out_v,
                       float* dh, float* d_lin_w, float* d_lin_b, float* d_v0_w, float* d_v0_b, float* d_v1_w,
                       float* d_v1_b, void* workspace, size_t workspace_bytes, hipStream_t st);

struct HeadSaved { size_t adv_off, pooled_off, amax_off, amin_off, z_off, v_off, total; };
HeadSaved head_saved_plan(int n, int b, int hidden);
struct HeadWs { size_t dadv_off, dz_off, dvr_off, part_off, total; };
HeadWs head_ws_plan(int n, int b, int hidden);
// value-head + advantage-linear parameter gradients from the per-graph partials
int launch_head_param_grads(int b, int hidden, int mode, const float* dz, const float* dvr, const float* pooled,
                            const float* z, const float* lin_part, float* d_lin_w, float* d_lin_b, float* d_v0_w,
                            float* d_v0_b, float* d_v1_w, float* d_v1_b, hipStream_t st);

}  // namespace hexgnn
