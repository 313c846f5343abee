// C ABI of the fused per-graph path + the exact-fp32 instantiations (kernels: qnet_fused_kernels.h).
#include "qnet_fused_kernels.h"

namespace hexgnn {

#define HEXGNN_NT_SWITCH7(nt, CALL)                     \
    switch (nt) {                                      \
        case 1: { constexpr int NT_ = 1; CALL; } break; \
        case 2: { constexpr int NT_ = 2; CALL; } break; \
        case 3: { constexpr int NT_ = 3; CALL; } break; \
        case 4: { constexpr int NT_ = 4; CALL; } break; \
        case 5: { constexpr int NT_ = 5; CALL; } break; \
        case 6: { constexpr int NT_ = 6; CALL; } break; \
        case 7: { constexpr int NT_ = 7; CALL; } break; \
        default: return HEXGNN_EUNSUPPORTED;           \
    }

int launch_qfwd_math(int nt, int math, const QFwdArgs& a, hipStream_t st) {
    if (math == 1) return launch_qfwd_split(nt, a, st);
    HEXGNN_NT_SWITCH7(nt, (launch_qfwd_m<NT_, 0>(a, st)));
    return HEXGNN_OK;
}
int launch_qbwd_math(int nt, int math, const QBwdArgs& a, hipStream_t st) {
    if (math == 1) return launch_qbwd_split(nt, a, st);
    HEXGNN_NT_SWITCH7(nt, (launch_qbwd_m<NT_, 0>(a, st)));
    return HEXGNN_OK;
}

}  // namespace hexgnn

using namespace hexgnn;

namespace {
struct QPlan {
    StackPlan sp;
    HeadSaved hs;
    size_t head_saved_off, xmax_off, saved_total;
    BwdPlan bp;
    HeadWs hw;
    size_t ws_g_off, ws_part_off, ws_part0_off, ws_head_off, ws_gmax_off, ws_total;
};
int make_qplan(int n, int b, int c_in, int hidden, int L, QPlan* q) {
    int rc = make_plan(n, c_in, hidden, L, &q->sp);
    if (rc != HEXGNN_OK) return rc;
    if (!q->sp.small_first || q->sp.nt > 7 || hidden < 2) return HEXGNN_EUNSUPPORTED;
    q->hs = head_saved_plan(n, b, hidden);
    q->head_saved_off = align_up(q->sp.saved_bytes, 256);
    q->xmax_off = align_up(q->head_saved_off + q->hs.total, 256);     // per-layer maxima (math 1), kMaxLayers words
    q->saved_total = q->xmax_off + sizeof(unsigned) * kMaxLayers;
    make_bwd_plan(n, q->sp, &q->bp);
    q->hw = head_ws_plan(n, b, hidden);
    const size_t slab = align_up(sizeof(float) * (size_t)n * q->sp.hp, 256);
    size_t off = 0;
    q->ws_g_off = off; off += slab * L;
    q->ws_part_off = off; off += align_up(sizeof(float) * (size_t)L * q->bp.S * q->sp.hp * (2 * q->sp.hp + 1), 256);
    q->ws_part0_off = off; off += align_up(sizeof(float) * (size_t)q->bp.S0 * q->sp.hp * 17, 256);
    q->ws_head_off = off; off += q->hw.total;
    q->ws_gmax_off = align_up(off, 256); off = q->ws_gmax_off + sizeof(unsigned) * kMaxLayers;
    q->ws_total = off;
    return HEXGNN_OK;
}
}  // namespace

extern "C" {

int hexgnn_qnet_supported(int c_in, int hidden, int max_nodes_per_graph) {
    const int hp = padded_width(hidden);
    return hp > 0 && hp <= 112 && hidden >= 2 && c_in >= 1 && c_in <= kSmallCin && c_in != hidden &&
           max_nodes_per_graph <= kRows;
}

size_t hexgnn_qnet_saved_bytes(int n, int b, int c_in, int hidden, int total_layers) {
    QPlan q;
    if (n < 0 || b < 0 || make_qplan(n, b, c_in, hidden, total_layers, &q) != HEXGNN_OK) return 0;
    return q.saved_total;
}

int hexgnn_qnet_forward(int n, int b, int c_in, int hidden, int total_layers, int mode, const int* gptr,
                        const int* rowptr, const int* col, const float* invdeg, const float* x, int x_stride,
                        const float* const* wl, const float* const* bl, const float* const* wr,
                        const float* lin_w, const float* lin_b, const float* v0_w, const float* v0_b,
                        const float* v1_w, const float* v1_b, void* wpack, float* acts, void* saved,
                        int need_backward, int math, float* q, float* out_v, int* status, hexgnn_stream_t stream_) {
    hipStream_t st = (hipStream_t)stream_;
    if (n < 0 || b < 0 || mode < 0 || mode > 2 || math < 0 || math > 1) return HEXGNN_EINVAL;
    QPlan qp;
    int rc = make_qplan(n, b, c_in, hidden, total_layers, &qp);
    if (rc != HEXGNN_OK) return rc;
    if (!gptr || !wl || !bl || !wr || !lin_w || !lin_b || !wpack || !saved || !status) return HEXGNN_EINVAL;
    if (mode != 2 && (!v0_w || !v0_b || !v1_w || !v1_b)) return HEXGNN_EINVAL;
    if (mode == 1 && !out_v) return HEXGNN_EINVAL;
    if (n > 0 && (!rowptr || !col || !invdeg || !x || !acts || !q)) return HEXGNN_EINVAL;
    if (x_stride < c_in) return HEXGNN_EINVAL;
    rc = launch_pack(qp.sp, c_in, hidden, wl, bl, wr, wpack, st, math);
    if (rc != HEXGNN_OK) return rc;
    if (b == 0) return check_launch();
    QFwdArgs a;
    a.n = n; a.b = b; a.c_in = c_in; a.H = hidden; a.L = total_layers; a.mode = mode; a.x_stride = x_stride;
    a.need_backward = need_backward;
    a.gptr = gptr; a.rowptr = rowptr; a.col = col; a.invdeg = invdeg; a.x = x;
    a.wpack = (const char*)wpack;
    for (int l = 0; l < total_layers; ++l) {
        a.fwd_off[l] = qp.sp.fwd_off[l]; a.bias_off[l] = qp.sp.bias_off[l]; a.agg_off[l] = qp.sp.agg_off[l];
    }
    a.acts = acts; a.saved = (char*)saved;
    a.lin_w = lin_w; a.lin_b = lin_b; a.v0_w = v0_w; a.v0_b = v0_b; a.v1_w = v1_w; a.v1_b = v1_b;
    char* hsv = (char*)saved + qp.head_saved_off;
    a.adv_raw = (float*)(hsv + qp.hs.adv_off); a.pooled = (float*)(hsv + qp.hs.pooled_off);
    a.amax = (int*)(hsv + qp.hs.amax_off); a.amin = (int*)(hsv + qp.hs.amin_off);
    a.z = (float*)(hsv + qp.hs.z_off); a.vraw = (float*)(hsv + qp.hs.v_off);
    a.q = q; a.out_v = out_v; a.status = status;
    a.xmax = nullptr;
    if (math == 1 && need_backward) {
        a.xmax = (unsigned*)((char*)saved + qp.xmax_off);
        (void)hipMemsetAsync(a.xmax, 0, sizeof(unsigned) * kMaxLayers, st);
    }
    {
        KernelTimer kt(HEXGNN_K_QNET_FWD, st);
        rc = launch_qfwd_math(qp.sp.nt, math, a, st);
        if (rc != HEXGNN_OK) return rc;
    }
    return check_launch();
}

size_t hexgnn_qnet_backward_workspace_bytes(int n, int b, int c_in, int hidden, int total_layers) {
    QPlan q;
    if (n < 0 || b < 0 || make_qplan(n, b, c_in, hidden, total_layers, &q) != HEXGNN_OK) return 0;
    return q.ws_total;
}

int hexgnn_qnet_backward(int n, int b, int c_in, int hidden, int total_layers, int body_layers, int mode, int math,
                         const int* gptr, const int* rowptr_t, const int* col_t, const float* invdeg,
                         const float* x, int x_stride, const float* acts, const void* saved, const void* wpack,
                         const float* lin_w, const float* v0_w, const float* v1_w, const float* dq,
                         const float* d_out_v, float* d_embeds, float* const* d_wl, float* const* d_bl,
                         float* const* d_wr, float* d_lin_w, float* d_lin_b, float* d_v0_w, float* d_v0_b,
                         float* d_v1_w, float* d_v1_b, void* workspace, size_t workspace_bytes, int* status,
                         hexgnn_stream_t stream_) {
    hipStream_t st = (hipStream_t)stream_;
    if (n < 0 || b < 0 || mode < 0 || mode > 2 || math < 0 || math > 1 || body_layers < 1 || body_layers > total_layers)
        return HEXGNN_EINVAL;
    QPlan qp;
    int rc = make_qplan(n, b, c_in, hidden, total_layers, &qp);
    if (rc != HEXGNN_OK) return rc;
    if (!workspace || workspace_bytes < qp.ws_total) return HEXGNN_EWORKSPACE;
    if (!gptr || !d_wl || !d_bl || !d_wr || !wpack || !saved || !lin_w || !d_lin_w || !d_lin_b || !status)
        return HEXGNN_EINVAL;
    for (int l = 0; l < total_layers; ++l) if (!d_wl[l] || !d_bl[l] || !d_wr[l]) return HEXGNN_EINVAL;
    if (mode != 2 && (!v0_w || !v1_w || !d_v0_w || !d_v0_b || !d_v1_w || !d_v1_b)) return HEXGNN_EINVAL;
    if (mode == 1 && !d_out_v) return HEXGNN_EINVAL;
    if (n > 0 && (!rowptr_t || !col_t || !invdeg || !x || !acts || !dq)) return HEXGNN_EINVAL;
    char* ws = (char*)workspace;
    const char* sv = (const char*)saved;
    const char* hsv = sv + qp.head_saved_off;
    float* G = (float*)(ws + qp.ws_g_off);
    float* part = (float*)(ws + qp.ws_part_off);
    float* part0 = (float*)(ws + qp.ws_part0_off);
    char* hws = ws + qp.ws_head_off;
    QBwdArgs a;
    a.n = n; a.b = b; a.H = hidden; a.L = total_layers; a.mode = mode; a.body_layers = body_layers;
    a.gptr = gptr; a.rowptr_t = rowptr_t; a.col_t = col_t; a.invdeg = invdeg;
    a.wpack = (const char*)wpack;
    for (int l = 0; l < total_layers; ++l) { a.bwd_off[l] = qp.sp.bwd_off[l]; a.bias_off[l] = qp.sp.bias_off[l]; }
    a.acts = acts; a.lin_w = lin_w; a.v0_w = v0_w; a.v1_w = v1_w;
    a.adv_raw = (const float*)(hsv + qp.hs.adv_off); a.amax = (const int*)(hsv + qp.hs.amax_off);
    a.amin = (const int*)(hsv + qp.hs.amin_off); a.z = (const float*)(hsv + qp.hs.z_off);
    a.vraw = (const float*)(hsv + qp.hs.v_off);
    a.dq = dq; a.d_out_v = d_out_v; a.G = G; a.d_embeds = d_embeds;
    a.dadv = (float*)(hws + qp.hw.dadv_off); a.dz = (float*)(hws + qp.hw.dz_off);
    a.dvr = (float*)(hws + qp.hw.dvr_off); a.lin_part = (float*)(hws + qp.hw.part_off);
    a.status = status;
    a.gmax = nullptr;
    if (math == 1) {
        a.gmax = (unsigned*)(ws + qp.ws_gmax_off);
        (void)hipMemsetAsync(a.gmax, 0, sizeof(unsigned) * kMaxLayers, st);
    }
    if (b > 0 && n > 0) {
        KernelTimer kt(HEXGNN_K_QNET_BWD, st);
        rc = launch_qbwd_math(qp.sp.nt, math, a, st);
        if (rc != HEXGNN_OK) return rc;
    } else {
        for (int l = 0; l < total_layers; ++l) {
            const int in = (l == 0) ? c_in : hidden;
            (void)hipMemsetAsync(d_wl[l], 0, sizeof(float) * (size_t)hidden * in, st);
            (void)hipMemsetAsync(d_wr[l], 0, sizeof(float) * (size_t)hidden * in, st);
            (void)hipMemsetAsync(d_bl[l], 0, sizeof(float) * (size_t)hidden, st);
        }
        (void)hipMemsetAsync(a.lin_part, 0, sizeof(float) * (size_t)(b > 0 ? b : 1) * (qp.sp.hp + 1), st);
        (void)hipMemsetAsync(a.dz, 0, sizeof(float) * (size_t)(b > 0 ? b : 1) * (hidden / 2), st);
        (void)hipMemsetAsync(a.dvr, 0, sizeof(float) * (size_t)(b > 0 ? b : 1), st);
    }
    if (n > 0) {
        rc = launch_weight_grads(n, c_in, hidden, qp.sp, qp.bp, x, x_stride, acts, sv, G, d_wl, d_bl, d_wr, part,
                                 part0, st, math, (const unsigned*)(sv + qp.xmax_off), a.gmax);
        if (rc != HEXGNN_OK) return rc;
    }
    launch_head_param_grads(b, hidden, mode, a.dz, a.dvr, (const float*)(hsv + qp.hs.pooled_off),
                            (const float*)(hsv + qp.hs.z_off), a.lin_part, d_lin_w, d_lin_b, d_v0_w, d_v0_b, d_v1_w,
                            d_v1_b, st);
    return check_launch();
}

}  // extern "C"
