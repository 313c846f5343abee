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


// ---- ONE launch for every reduction that turns the backward's partial results into parameter gradients --------------
// roles by block range (256 threads each):
//   R0  hidden-layer dW / db: sum of the S row-slice slabs written by sage_dw(16)_kernel            (fixed order over s)
//   R1  raw first layer dW / db: sum over graphs of the per-graph partials of qnet_bwd_kernel        (wave per output)
//   R2  advantage Linear: sum over graphs of lin_part                                               (wave per output)
//   R3  value MLP: d_v0_w = dz^T pooled, d_v0_b, d_v1_w, d_v1_b                                     (as head_value_wgrad_kernel)
// Every sum has a fixed shape => bit-reproducible.  Replaces five launches of the layered path.
struct GradReduceArgs {
    float* dwl[kMaxLayers]; float* dbl[kMaxLayers]; float* dwr[kMaxLayers];   // hidden layers (index = hidden layer)
    const float* part; int S, hp, H, nh, blk_per_layer, per;
    const float* first_part; int b, c_in; float* dwl0; float* dbl0; float* dwr0;
    const float* lin_part; float* d_lin_w; float* d_lin_b;
    const float* dz; const float* dvr; const float* pooled; const float* z;
    float* d_v0_w; float* d_v0_b; float* d_v1_w; float* d_v1_b;
    int n0, n1, n2, n3, nbx3;
    // R4 (one block, hexgnn_qnet_backward_flat_td): loss = (sum over the b graphs of loss_part) / b, in td_loss_fwd_kernel's
    // reduction shape (thread t sums entries t, t + 256, ...; tree over the 256 threads): its bits
    const float* loss_part; float* loss;
};

__device__ __forceinline__ float wsum_all(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__global__ __launch_bounds__(256) void qnet_grad_reduce_kernel(GradReduceArgs a) {
    int blk = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = a.H, hp = a.hp;
    if (blk == a.n0 + a.n1 + a.n2 + a.n3) {      // ---- R4: the TD loss's mean over the graphs
        __shared__ float lred[256];
        float acc = 0.f;
        for (int j = tid; j < a.b; j += 256) acc += a.loss_part[j];
        lred[tid] = acc;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) lred[tid] += lred[tid + o];
            __syncthreads();
        }
        if (tid == 0) a.loss[0] = lred[0] / (float)(a.b > 0 ? a.b : 1);
        return;
    }
    if (blk < a.n0) {                     // ---- R0: slice slabs -> dW_l / dW_r / db of one hidden layer
        // one thread per float4 of the [hp][2hp] slab (+ the bias row), S independent 16-byte loads in flight
        // (blk_per_layer x hidden layers ~ two workgroups per CU, every one with the same share `per` of the slab: 400 blocks
        // of 256 float4 left 144 CUs with twice the bytes of the other 112, and the kernel is bound by bytes per CU)
        const int li = blk / a.blk_per_layer, idx = (blk % a.blk_per_layer) * a.per + tid;
        const int q_row = 2 * hp / 4, q_w = hp * q_row, q_all = q_w + hp / 4;
        if (tid >= a.per || idx >= q_all) return;
        const size_t slab_sz = (size_t)hp * (2 * hp + 1);
        const f32x4* p = reinterpret_cast<const f32x4*>(a.part + (size_t)li * a.S * slab_sz) + idx;   // slab_sz % 4 == 0
        f32x4 sum = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
        for (int s = 0; s < a.S; ++s) sum += p[(size_t)s * (slab_sz / 4)];
        if (idx < q_w) {
            const int o = idx / q_row, c0 = (idx % q_row) * 4;
            if (o >= H) return;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = c0 + j;
                if (c < H) a.dwl[li][o * H + c] = sum[j];
                else if (c >= hp && c - hp < H) a.dwr[li][o * H + (c - hp)] = sum[j];
            }
        } else {
            const int o0 = (idx - q_w) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) if (o0 + j < H) a.dbl[li][o0 + j] = sum[j];
        }
        return;
    }
    blk -= a.n0;
    if (blk < a.n1) {                     // ---- R1
        const int per = 2 * a.c_in + 1;
        const int idx = blk * 4 + wave;
        if (idx >= H * per) return;
        const int o = idx / per, c = idx % per;
        const int q = c < a.c_in ? c : (c < 2 * a.c_in ? kSmallCin + (c - a.c_in) : 2 * kSmallCin);
        float sum = 0.f;
        for (int g = lane; g < a.b; g += 64) sum += a.first_part[((size_t)g * 17 + q) * hp + o];
        sum = wsum_all(sum);
        if (lane == 0) {
            if (c < a.c_in) a.dwl0[o * a.c_in + c] = sum;
            else if (c < 2 * a.c_in) a.dwr0[o * a.c_in + (c - a.c_in)] = sum;
            else a.dbl0[o] = sum;
        }
        return;
    }
    blk -= a.n1;
    if (blk < a.n2) {                     // ---- R2
        const int c = blk * 4 + wave;     // c in [0, H]  (H == bias)
        if (c > H) return;
        const int src = c < H ? c : hp;
        float s = 0.f;
        for (int g = lane; g < a.b; g += 64) s += a.lin_part[(size_t)g * (hp + 1) + src];
        s = wsum_all(s);
        if (lane == 0) { if (c < H) a.d_lin_w[c] = s; else a.d_lin_b[0] = s; }
        return;
    }
    blk -= a.n2;
    {                                     // ---- R3
        // 16 columns x 16 graph phases per block (round 4): with 64 columns x 4 phases a thread walked 64 graphs one dependent
        // batch of loads after the other, and these blocks -- not the slab sums -- set the length of the launch
        const int H2 = H / 2, H4 = 4 * H;
        const int bx = blk % a.nbx3, k = blk / a.nbx3;
        const int cl = tid & 15, ph = tid >> 4;
        const int c = bx * 16 + cl;
        __shared__ float red16[16][17];
        float s = 0.f;
        if (c < H4) {
#pragma unroll 8
            for (int g = ph; g < a.b; g += 16) s += a.dz[(size_t)g * H2 + k] * a.pooled[(size_t)g * H4 + c];
        }
        red16[ph][cl] = s;
        __syncthreads();
        if (ph == 0 && c < H4) {
            float t = red16[0][cl];
#pragma unroll
            for (int j = 1; j < 16; ++j) t += red16[j][cl];          // fixed order: deterministic
            a.d_v0_w[(size_t)k * H4 + c] = t;
        }
        if (bx == 0) {
            float p = 0.f;
            if (wave == 0) { for (int g = lane; g < a.b; g += 64) p += a.dz[(size_t)g * H2 + k]; }
            else if (wave == 1) { for (int g = lane; g < a.b; g += 64) p += a.dvr[g] * a.z[(size_t)g * H2 + k]; }
            else if (wave == 2 && k == 0) { for (int g = lane; g < a.b; g += 64) p += a.dvr[g]; }
            p = wsum_all(p);
            if (lane == 0) {
                if (wave == 0) a.d_v0_b[k] = p;
                else if (wave == 1) a.d_v1_w[k] = p;
                else if (wave == 2 && k == 0) a.d_v1_b[0] = p;
            }
        }
    }
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
    size_t ws_g_off, ws_part_off, ws_part0_off, ws_head_off, ws_first_off, ws_total;
};
int make_qplan(int n, int b, int c_in, int hidden, int L, QPlan* q) {
    int rc = make_plan(n, c_in, hidden, L, &q->sp);
    if (rc != HEXGNN_OK) return rc;
    if (!q->sp.small_first || q->sp.nt > 7 || hidden < 2) return HEXGNN_EUNSUPPORTED;
    q->hs = head_saved_plan(n, b, hidden);
    q->head_saved_off = align_up(q->sp.saved_bytes, 256);
    q->xmax_off = align_up(q->head_saved_off + q->hs.total, 256);     // per-layer maxima (math 1), kMaxLayers words
    q->saved_total = q->xmax_off + 2 * sizeof(unsigned) * kMaxLayers;   // xmax[kMaxLayers] then gmax[kMaxLayers]
    make_bwd_plan(n, q->sp, &q->bp);
    q->hw = head_ws_plan(n, b, hidden);
    const size_t slab = align_up(sizeof(float) * (size_t)n * q->sp.hp, 256);
    size_t off = 0;
    q->ws_g_off = off; off += slab * L;
    q->ws_part_off = off; off += align_up(sizeof(float) * dw_slab_count(L) * q->sp.hp * (2 * q->sp.hp + 1), 256);
    q->ws_part0_off = off; off += align_up(sizeof(float) * (size_t)q->bp.S0 * q->sp.hp * 17, 256);
    q->ws_head_off = off; off += q->hw.total;
    q->ws_first_off = align_up(off, 256); off = q->ws_first_off + sizeof(float) * (size_t)(b > 0 ? b : 1) * 17 * q->sp.hp;
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

static int qnet_forward_impl(int n, int b, int c_in, int hidden, int total_layers, int mode, const int* gptr,
                             const int* rowptr, const int* col, const float* invdeg, const float* x, int x_stride,
                             const float* const* wl, const float* const* bl, const float* const* wr,
                             const float* lin_w, const float* lin_b, const float* v0_w, const float* v0_b,
                             const float* v1_w, const float* v1_b, void* wpack, float* acts, void* saved,
                             int need_backward, int acts_layer, int math, float* q, float* out_v, int* status,
                             const int64_t* td_sel, const float* td_target, const float* td_weights, int td_loss_fn,
                             float* td_dq, float* td_out, float* td_loss_part, hexgnn_stream_t stream_) {
    hipStream_t st = (hipStream_t)stream_;
    if (n < 0 || b < 0 || mode < 0 || mode > 2 || math < 0 || math > 1) return HEXGNN_EINVAL;
    QPlan qp;
    int rc = make_qplan(n, b, c_in, hidden, total_layers, &qp);
    if (rc != HEXGNN_OK) return rc;
    // (wl = bl = wr = NULL: the weights were packed by hexgnn_csr_build_grouped_pack of this forward; exact fp32 only)
    const bool packed = !wl && !bl && !wr && math == 0;
    if (!gptr || (!packed && (!wl || !bl || !wr)) || !lin_w || !lin_b || !wpack || !saved || !status) return HEXGNN_EINVAL;
    if (mode != 2 && (!v0_w || !v0_b || !v1_w || !v1_b)) return HEXGNN_EINVAL;
    if (mode == 1 && !out_v) return HEXGNN_EINVAL;
    if (n > 0 && (!rowptr || !col || !invdeg || !x || !acts || !q)) return HEXGNN_EINVAL;
    if (x_stride < c_in) return HEXGNN_EINVAL;
    // math 1 + backward: the pack's scale kernel also zeroes the per-layer maxima (xmax | gmax) kept behind the saved tensors
    unsigned* maxima = (math == 1 && need_backward) ? (unsigned*)((char*)saved + qp.xmax_off) : nullptr;
    rc = launch_pack(qp.sp, c_in, hidden, wl, bl, wr, wpack, st, math, maxima);
    if (rc != HEXGNN_OK) return rc;
    if (b == 0) return check_launch();
    QFwdArgs a;
    a.n = n; a.b = b; a.c_in = c_in; a.H = hidden; a.L = total_layers; a.mode = mode; a.x_stride = x_stride;
    a.need_backward = need_backward;
    a.acts_layer = acts_layer;
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
    a.xmax = maxima;
    a.td_sel = (const long long*)td_sel; a.td_tgt = td_target; a.td_w = td_weights; a.td_loss_fn = td_loss_fn;
    a.td_dq = td_dq; a.td_out = td_out; a.td_loss_part = td_loss_part;
    {
        KernelTimer kt(HEXGNN_K_QNET_FWD, st);
        rc = launch_qfwd_math(qp.sp.nt, math, a, st);
        if (rc != HEXGNN_OK) return rc;
    }
    return check_launch();
}

int hexgnn_qnet_forward(int n, int b, int c_in, int hidden, int total_layers, int mode, const int* gptr,
                        const int* rowptr, const int* col, const float* invdeg, const float* x, int x_stride,
                        const float* const* wl, const float* const* bl, const float* const* wr,
                        const float* lin_w, const float* lin_b, const float* v0_w, const float* v0_b,
                        const float* v1_w, const float* v1_b, void* wpack, float* acts, void* saved,
                        int need_backward, int acts_layer, int math, float* q, float* out_v, int* status,
                        hexgnn_stream_t stream_) {
    return qnet_forward_impl(n, b, c_in, hidden, total_layers, mode, gptr, rowptr, col, invdeg, x, x_stride, wl, bl, wr, lin_w,
                             lin_b, v0_w, v0_b, v1_w, v1_b, wpack, acts, saved, need_backward, acts_layer, math, q, out_v,
                             status, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, stream_);
}

int hexgnn_qnet_forward_td(int n, int b, int c_in, int hidden, int total_layers, const int* gptr,
                           const int* rowptr, const int* col, const float* invdeg, const float* x, int x_stride,
                           const float* const* wl, const float* const* bl, const float* const* wr,
                           const float* lin_w, const float* lin_b, const float* v0_w, const float* v0_b,
                           const float* v1_w, const float* v1_b, void* wpack, float* acts, void* saved,
                           int math, float* q, int* status, const int64_t* sel, const float* target,
                           const float* weights, int loss_fn, float* dq, float* td, float* loss_part,
                           hexgnn_stream_t stream_) {
    if (!sel || !target || !dq || !td || !loss_part || loss_fn < 0 || loss_fn > 1) return HEXGNN_EINVAL;
    return qnet_forward_impl(n, b, c_in, hidden, total_layers, 0, gptr, rowptr, col, invdeg, x, x_stride, wl, bl, wr, lin_w,
                             lin_b, v0_w, v0_b, v1_w, v1_b, wpack, acts, saved, 1, -1, math, q, nullptr, status, sel, target,
                             weights, loss_fn, dq, td, loss_part, stream_);
}

size_t hexgnn_qnet_backward_workspace_bytes(int n, int b, int c_in, int hidden, int total_layers) {
    QPlan q;
    if (n < 0 || b < 0 || make_qplan(n, b, c_in, hidden, total_layers, &q) != HEXGNN_OK) return 0;
    return q.ws_total;
}

static int qnet_backward_staged_impl(int n, int b, int c_in, int hidden, int total_layers, int body_layers, int mode, int math,
                                const int* gptr, const int* rowptr_t, const int* col_t, const float* invdeg,
                                const float* x, int x_stride, const float* acts, const void* saved, const void* wpack,
                                const float* lin_w, const float* v0_w, const float* v1_w, const float* dq,
                                const float* d_out_v, float* d_embeds, float* const* d_wl, float* const* d_bl,
                                float* const* d_wr, float* d_lin_w, float* d_lin_b, float* d_v0_w, float* d_v0_b,
                                float* d_v1_w, float* d_v1_b, void* workspace, size_t workspace_bytes, int* status,
                                int stages, int layer_lo, int layer_hi, const float* loss_part, float* loss,
                                hexgnn_stream_t stream_) {
    hipStream_t st = (hipStream_t)stream_;
    if (n < 0 || b < 0 || mode < 0 || mode > 2 || math < 0 || math > 1 || body_layers < 1 || body_layers > total_layers)
        return HEXGNN_EINVAL;
    if (stages <= 0 || (stages & ~(HEXGNN_QBWD_DATA | HEXGNN_QBWD_SMALL | HEXGNN_QBWD_HIDDEN))) return HEXGNN_EINVAL;
    if ((stages & HEXGNN_QBWD_HIDDEN) && (layer_lo < 1 || layer_hi < layer_lo || layer_hi > total_layers)) return HEXGNN_EINVAL;
    const bool st_data = stages & HEXGNN_QBWD_DATA, st_small = stages & HEXGNN_QBWD_SMALL, st_hidden = stages & HEXGNN_QBWD_HIDDEN;
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
    a.x = x; a.x_stride = x_stride; a.c_in = c_in;
    a.agg0 = (const float*)(sv + qp.sp.agg_off[0]);
    a.first_part = (float*)(ws + qp.ws_first_off);
    // max |G_l| per layer (math 1): lives behind the forward's saved tensors, zeroed by the forward; a backward pass
    // repeated on the same saved state only re-maxes identical values
    a.gmax = math == 1 ? (unsigned*)(const_cast<char*>(sv) + qp.xmax_off) + kMaxLayers : nullptr;
    if (b > 0 && n > 0) {
        if (st_data) {
            KernelTimer kt(HEXGNN_K_QNET_BWD, st);
            rc = launch_qbwd_math(qp.sp.nt, math, a, st);
            if (rc != HEXGNN_OK) return rc;
        }
    } else if (st_data) {
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
    if (n > 0 && b > 0) {
        // hidden layers [lo, hi) of this call: weight-gradient GEMM into the stage's own slab region, then ONE reduce launch
        // whose block roles cover those slabs (R0) and -- with HEXGNN_QBWD_SMALL -- the per-graph partials (R1..R3)
        const int lo = st_hidden ? layer_lo : 1, hi = st_hidden ? layer_hi : 1;
        const int nh = hi - lo;
        const size_t slab_sz = (size_t)qp.sp.hp * (2 * qp.sp.hp + 1);
        float* spart = part + (size_t)(lo - 1) * kDwMaxSlices * slab_sz;
        if (nh > 0) {
            rc = launch_weight_grads(n, c_in, hidden, qp.sp, qp.bp, x, x_stride, acts, sv, G, d_wl, d_bl, d_wr, spart,
                                     part0, st, math, (const unsigned*)(sv + qp.xmax_off), a.gmax,
                                     /*hidden_only_no_reduce=*/true, lo, hi);
            if (rc != HEXGNN_OK) return rc;
        }
        GradReduceArgs r;
        for (int i = 0; i < nh; ++i) { r.dwl[i] = d_wl[lo + i]; r.dbl[i] = d_bl[lo + i]; r.dwr[i] = d_wr[lo + i]; }
        r.part = spart; r.S = dw_slices_for(n, nh, math, qp.sp.L - (qp.sp.small_first ? 1 : 0)); r.hp = qp.sp.hp; r.H = hidden; r.nh = nh;
        {   // one thread per float4 of a slab; blocks per layer so that all layers together make ~512 equal blocks
            const int q_all = qp.sp.hp * (2 * qp.sp.hp + 1) / 4;
            int bpl = (q_all + 255) / 256;
            if (nh > 0 && bpl * nh < 512) bpl = (512 + nh - 1) / nh;
            if (bpl > (q_all + 31) / 32) bpl = (q_all + 31) / 32;
            r.blk_per_layer = bpl;
            r.per = (q_all + bpl - 1) / bpl;
        }
        r.first_part = a.first_part; r.b = b; r.c_in = c_in; r.dwl0 = d_wl[0]; r.dbl0 = d_bl[0]; r.dwr0 = d_wr[0];
        r.lin_part = a.lin_part; r.d_lin_w = d_lin_w; r.d_lin_b = d_lin_b;
        r.dz = a.dz; r.dvr = a.dvr; r.pooled = (const float*)(hsv + qp.hs.pooled_off); r.z = (const float*)(hsv + qp.hs.z_off);
        r.d_v0_w = d_v0_w; r.d_v0_b = d_v0_b; r.d_v1_w = d_v1_w; r.d_v1_b = d_v1_b;
        r.n0 = nh * r.blk_per_layer;
        r.n1 = st_small ? (hidden * (2 * c_in + 1) + 3) / 4 : 0;
        r.n2 = st_small ? (hidden + 1 + 3) / 4 : 0;
        r.nbx3 = (4 * hidden + 15) / 16;
        r.n3 = (st_small && mode != 2 && hidden / 2 > 0) ? r.nbx3 * (hidden / 2) : 0;
        r.loss_part = loss_part; r.loss = loss;
        const int n4 = (st_small && loss_part && loss) ? 1 : 0;      // (the block behind the last role: the TD loss's mean)
        if (r.n0 + r.n1 + r.n2 + r.n3 + n4 > 0) qnet_grad_reduce_kernel<<<r.n0 + r.n1 + r.n2 + r.n3 + n4, 256, 0, st>>>(r);
    } else if (st_small) {
        launch_head_param_grads(b, hidden, mode, a.dz, a.dvr, (const float*)(hsv + qp.hs.pooled_off),
                                (const float*)(hsv + qp.hs.z_off), a.lin_part, d_lin_w, d_lin_b, d_v0_w, d_v0_b, d_v1_w,
                                d_v1_b, st);
    }
    return check_launch();
}

int hexgnn_qnet_backward_staged(int n, int b, int c_in, int hidden, int total_layers, int body_layers, int mode, int math,
                                const int* gptr, const int* rowptr_t, const int* col_t, const float* invdeg,
                                const float* x, int x_stride, const float* acts, const void* saved, const void* wpack,
                                const float* lin_w, const float* v0_w, const float* v1_w, const float* dq,
                                const float* d_out_v, float* d_embeds, float* const* d_wl, float* const* d_bl,
                                float* const* d_wr, float* d_lin_w, float* d_lin_b, float* d_v0_w, float* d_v0_b,
                                float* d_v1_w, float* d_v1_b, void* workspace, size_t workspace_bytes, int* status,
                                int stages, int layer_lo, int layer_hi, hexgnn_stream_t stream_) {
    return qnet_backward_staged_impl(n, b, c_in, hidden, total_layers, body_layers, mode, math, gptr, rowptr_t, col_t, invdeg,
                                     x, x_stride, acts, saved, wpack, lin_w, v0_w, v1_w, dq, d_out_v, d_embeds, d_wl, d_bl,
                                     d_wr, d_lin_w, d_lin_b, d_v0_w, d_v0_b, d_v1_w, d_v1_b, workspace, workspace_bytes,
                                     status, stages, layer_lo, layer_hi, nullptr, nullptr, stream_);
}

int hexgnn_qnet_backward(int n, int b, int c_in, int hidden, int total_layers, int body_layers, int mode, int math,
                         const int* gptr, const int* rowptr_t, const int* col_t, const float* invdeg,
                         const float* x, int x_stride, const float* acts, const void* saved, const void* wpack,
                         const float* lin_w, const float* v0_w, const float* v1_w, const float* dq,
                         const float* d_out_v, float* d_embeds, float* const* d_wl, float* const* d_bl,
                         float* const* d_wr, float* d_lin_w, float* d_lin_b, float* d_v0_w, float* d_v0_b,
                         float* d_v1_w, float* d_v1_b, void* workspace, size_t workspace_bytes, int* status,
                         hexgnn_stream_t stream_) {
    return hexgnn_qnet_backward_staged(n, b, c_in, hidden, total_layers, body_layers, mode, math, gptr, rowptr_t, col_t,
                                       invdeg, x, x_stride, acts, saved, wpack, lin_w, v0_w, v1_w, dq, d_out_v, d_embeds,
                                       d_wl, d_bl, d_wr, d_lin_w, d_lin_b, d_v0_w, d_v0_b, d_v1_w, d_v1_b, workspace,
                                       workspace_bytes, status, HEXGNN_QBWD_DATA | HEXGNN_QBWD_SMALL | HEXGNN_QBWD_HIDDEN,
                                       1, total_layers, stream_);
}

int hexgnn_qnet_backward_flat(int n, int b, int c_in, int hidden, int total_layers, int body_layers, int mode, int math,
                              const int* gptr, const int* rowptr_t, const int* col_t, const float* invdeg,
                              const float* x, int x_stride, const float* acts, const void* saved, const void* wpack,
                              const float* lin_w, const float* v0_w, const float* v1_w, const float* dq,
                              const float* d_out_v, float* d_embeds, float* flat, const int64_t* offsets,
                              void* workspace, size_t workspace_bytes, int* status, int stages, int layer_lo,
                              int layer_hi, hexgnn_stream_t stream_) {
    if (!flat || !offsets || total_layers < 1 || total_layers > kMaxLayers) return HEXGNN_EINVAL;
    float* d_wl[kMaxLayers]; float* d_bl[kMaxLayers]; float* d_wr[kMaxLayers];
    for (int l = 0; l < total_layers; ++l) {
        d_wl[l] = flat + offsets[3 * l]; d_bl[l] = flat + offsets[3 * l + 1]; d_wr[l] = flat + offsets[3 * l + 2];
    }
    const int64_t* t = offsets + 3 * total_layers;      // lin_w, lin_b, v0_w, v0_b, v1_w, v1_b
    const bool vh = mode != 2;
    return hexgnn_qnet_backward_staged(n, b, c_in, hidden, total_layers, body_layers, mode, math, gptr, rowptr_t, col_t,
                                       invdeg, x, x_stride, acts, saved, wpack, lin_w, v0_w, v1_w, dq, d_out_v, d_embeds,
                                       d_wl, d_bl, d_wr, flat + t[0], flat + t[1], vh ? flat + t[2] : nullptr,
                                       vh ? flat + t[3] : nullptr, vh ? flat + t[4] : nullptr, vh ? flat + t[5] : nullptr,
                                       workspace, workspace_bytes, status, stages, layer_lo, layer_hi, stream_);
}

int hexgnn_qnet_backward_flat_td(int n, int b, int c_in, int hidden, int total_layers, int body_layers, int math,
                                 const int* gptr, const int* rowptr_t, const int* col_t, const float* invdeg,
                                 const float* x, int x_stride, const float* acts, const void* saved, const void* wpack,
                                 const float* lin_w, const float* v0_w, const float* v1_w, const float* dq,
                                 float* d_embeds, float* flat, const int64_t* offsets, void* workspace,
                                 size_t workspace_bytes, int* status, int stages, int layer_lo, int layer_hi,
                                 const float* loss_part, float* loss, hexgnn_stream_t stream_) {
    if (!flat || !offsets || total_layers < 1 || total_layers > kMaxLayers || !loss_part || !loss) return HEXGNN_EINVAL;
    float* d_wl[kMaxLayers]; float* d_bl[kMaxLayers]; float* d_wr[kMaxLayers];
    for (int l = 0; l < total_layers; ++l) {
        d_wl[l] = flat + offsets[3 * l]; d_bl[l] = flat + offsets[3 * l + 1]; d_wr[l] = flat + offsets[3 * l + 2];
    }
    const int64_t* t = offsets + 3 * total_layers;      // lin_w, lin_b, v0_w, v0_b, v1_w, v1_b
    return qnet_backward_staged_impl(n, b, c_in, hidden, total_layers, body_layers, 0, math, gptr, rowptr_t, col_t, invdeg, x,
                                     x_stride, acts, saved, wpack, lin_w, v0_w, v1_w, dq, nullptr, d_embeds, d_wl, d_bl,
                                     d_wr, flat + t[0], flat + t[1], flat + t[2], flat + t[3], flat + t[4], flat + t[5],
                                     workspace, workspace_bytes, status, stages, layer_lo, layer_hi, loss_part, loss,
                                     stream_);
}

#ifdef HEXGNN_STAMPS
// profiling builds only: copies the s_memtime stamps of the exact-fp32 fused kernels to `out` (host pointer)
int hexgnn_debug_stamps(unsigned long long* out, int capacity) {
    const int total = 2 * (kMaxLayers + 2) * kStampPoints * 8;
    if (capacity < total) return HEXGNN_EINVAL;
    if (hipDeviceSynchronize() != hipSuccess) return HEXGNN_EHIP;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(hexgnn::g_qstamps), sizeof(unsigned long long) * total) != hipSuccess) return HEXGNN_EHIP;
    return total;
}
#endif
}  // extern "C"
