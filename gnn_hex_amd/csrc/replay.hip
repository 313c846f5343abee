// Prioritized experience replay on the device: proportional prioritisation (Schaul et al. 2016) over a sum tree and
// a min tree kept in HBM, stratified sampling, importance weights.
//
// Reference: the RainbowDQN replay buffer lives in the un-vendored submodule GN0/RainbowDQN/Rainbow (.gitmodules:1-4,
// fork of schmidtdominik/Rainbow); only its flags are visible (README.md:5,7: --prioritized_er=True
// --prioritized_er_beta0=0.6 --buffer_size=260000 --n_step=2).  PARITY UNPINNED: this follows the published
// segment-tree algorithm (OpenAI-baselines form: priorities p^alpha in a sum tree and a min tree of capacity 2^k;
// batch sample i draws mass = (i + u_i) * total / B and descends the sum tree; weight_i = (N * p_i / total)^-beta /
// (N * p_min / total)^-beta) and is checked bit for bit against oracle/replay_ref.py.
// Trees are fp64 (deterministic parent = left + right), node 1 is the root, leaves at [cap, 2cap).
#include "hexgnn_common.h"

namespace hexgnn {

// set leaves then rebuild the touched ancestors level by level; one workgroup (updates per call <= a few thousand).
// A slot listed more than once (PER samples with replacement, so update_priorities sees duplicates) takes the priority of
// its LAST occurrence, as a sequential loop would: an occurrence writes only if no later entry names the same slot
// (pairwise scan of the list in LDS; the host entry point feeds at most 2048 entries per launch), so sum and min tree
// always agree.
//
// Fused form (hexgnn_per_update_td): the priorities come from the caller's TD errors, p_i = |td_i| + eps, leaf value p_i^alpha,
// and the running maximum priority *max_prio (what new transitions are stored with) is raised to max_i p_i -- or, with
// td == nullptr, every listed slot gets (*max_prio)^alpha (a block of new transitions).  One launch instead of the ten small
// torch kernels of abs / cast / add / max / maximum / pow / index conversion.
struct PerFused {
    const float* td; const long long* idx64; double alpha, eps; double* max_prio;
};
__global__ __launch_bounds__(1024) void per_update_kernel(int cap, int k, const int* __restrict__ idx,
                                                         const double* __restrict__ prio_alpha, PerFused f,
                                                         double* __restrict__ sum_tree, double* __restrict__ min_tree) {
    __shared__ __attribute__((aligned(16))) int s_idx[2048];
    __shared__ double s_red[16];
    for (int i = threadIdx.x; i < 2048; i += 1024) {     // k <= 2048 (host entry point)
        int v = -1;
        if (i < k) {
            if (f.idx64) { const long long w = f.idx64[i]; v = (w >= 0 && w < (long long)cap) ? (int)w : -1; }
            else v = idx[i];
        }
        s_idx[i] = v;
    }
    const bool fused = prio_alpha == nullptr;
    double fill = 0.0;
    if (fused && f.td) {         // running maximum first (block reduce; one workgroup per launch, launches are stream-ordered)
        double m = 0.0;
        for (int i = threadIdx.x; i < k; i += 1024) { const double p = (double)fabsf(f.td[i]) + f.eps; m = p > m ? p : m; }
        for (int o = 32; o >= 1; o >>= 1) { const double t = __shfl_xor(m, o); m = t > m ? t : m; }
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            double mm = *f.max_prio;
            for (int w = 0; w < 16; ++w) mm = s_red[w] > mm ? s_red[w] : mm;
            *f.max_prio = mm;
        }
    } else if (fused) {
        fill = pow(*f.max_prio, f.alpha);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < k; i += 1024) {
        const int slot = s_idx[i];
        if (slot < 0 || slot >= cap) continue;               // out-of-range slots are ignored
        // a later entry with the same slot?  Four entries per LDS access, no early exit (a dependent exit test makes every
        // iteration wait for its own LDS round trip)
        bool last = true;
        for (int q4 = (i + 1) / 4; q4 < (k + 3) / 4; ++q4) {
            const int4 v = reinterpret_cast<const int4*>(s_idx)[q4];
            const int q = 4 * q4;
            last = last && !((v.x == slot && q > i) || (v.y == slot && q + 1 > i) || (v.z == slot && q + 2 > i) ||
                             (v.w == slot && q + 3 > i));
        }
        if (!last) continue;
        const int leaf = cap + slot;
        const double pa = !fused ? prio_alpha[i] : (f.td ? pow((double)fabsf(f.td[i]) + f.eps, f.alpha) : fill);
        sum_tree[leaf] = pa;
        min_tree[leaf] = pa;
    }
    __syncthreads();
    for (int width = cap >> 1, shift = 1; width >= 1; width >>= 1, ++shift) {
        for (int i = threadIdx.x; i < k; i += 1024) {
            const int slot = s_idx[i];
            if (slot < 0 || slot >= cap) continue;
            const int node = (cap + slot) >> shift;         // duplicates recompute the same value
            const double l = sum_tree[2 * node], r = sum_tree[2 * node + 1];
            sum_tree[node] = l + r;
            const double ml = min_tree[2 * node], mr = min_tree[2 * node + 1];
            min_tree[node] = ml < mr ? ml : mr;
        }
        __syncthreads();
    }
}

__global__ void per_sample_kernel(int cap, int size, int b, double beta, const double* __restrict__ u,
                                  const double* __restrict__ sum_tree, const double* __restrict__ min_tree,
                                  int* __restrict__ out_idx, float* __restrict__ out_w) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b) return;
    const double total = sum_tree[1];
    double mass = ((double)i + u[i]) * (total / (double)b);
    int node = 1;
    while (node < cap) {                                     // find_prefixsum_idx
        const double l = sum_tree[2 * node];
        if (l > mass) node = 2 * node;
        else { mass -= l; node = 2 * node + 1; }
    }
    int leaf = node - cap;
    if (leaf >= size) leaf = size - 1;                       // rounding at the right edge
    out_idx[i] = leaf;
    const double p_min = min_tree[1] / total;
    const double max_w = pow(p_min * (double)size, -beta);
    const double p = sum_tree[cap + leaf] / total;
    out_w[i] = (float)(pow(p * (double)size, -beta) / max_w);
}

__global__ void per_fill_kernel(int n, double v_sum, double v_min, double* __restrict__ sum_tree,
                                double* __restrict__ min_tree) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { sum_tree[i] = v_sum; min_tree[i] = v_min; }
}

}  // namespace hexgnn

using namespace hexgnn;

extern "C" {

int hexgnn_per_init(int capacity_pow2, double* sum_tree, double* min_tree, hexgnn_stream_t stream_) {
    if (capacity_pow2 < 1 || (capacity_pow2 & (capacity_pow2 - 1)) || !sum_tree || !min_tree) return HEXGNN_EINVAL;
    const int n = 2 * capacity_pow2;
    per_fill_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream_>>>(n, 0.0, INFINITY, sum_tree, min_tree);
    return check_launch();
}

int hexgnn_per_update(int capacity_pow2, int k, const int* idx, const double* prio_alpha, double* sum_tree,
                      double* min_tree, hexgnn_stream_t stream_) {
    if (capacity_pow2 < 1 || (capacity_pow2 & (capacity_pow2 - 1)) || k < 0 || !sum_tree || !min_tree) return HEXGNN_EINVAL;
    if (k == 0) return HEXGNN_OK;
    if (!idx || !prio_alpha) return HEXGNN_EINVAL;
    // chunks of <= 2048 entries in list order: the duplicate scan stays in LDS, and a slot repeated across chunks still ends
    // with its last occurrence (stream order)
    for (int o = 0; o < k; o += 2048) {
        const int kk = k - o < 2048 ? k - o : 2048;
        per_update_kernel<<<1, 1024, 0, (hipStream_t)stream_>>>(capacity_pow2, kk, idx + o, prio_alpha + o, PerFused{}, sum_tree, min_tree);
    }
    return check_launch();
}

int hexgnn_per_update_td(int capacity_pow2, int k, const void* idx, int idx_bits, const float* td, double alpha, double eps,
                         double* max_priority, double* sum_tree, double* min_tree, hexgnn_stream_t stream_) {
    if (capacity_pow2 < 1 || (capacity_pow2 & (capacity_pow2 - 1)) || k < 0 || !sum_tree || !min_tree || !max_priority ||
        (idx_bits != 32 && idx_bits != 64) || !(alpha >= 0.0) || !(eps >= 0.0))
        return HEXGNN_EINVAL;
    if (k == 0) return HEXGNN_OK;
    if (!idx) return HEXGNN_EINVAL;
    for (int o = 0; o < k; o += 2048) {
        const int kk = k - o < 2048 ? k - o : 2048;
        PerFused f{td ? td + o : nullptr, idx_bits == 64 ? static_cast<const long long*>(idx) + o : nullptr, alpha, eps,
                   max_priority};
        per_update_kernel<<<1, 1024, 0, (hipStream_t)stream_>>>(capacity_pow2, kk,
                                                                idx_bits == 32 ? static_cast<const int*>(idx) + o : nullptr,
                                                                nullptr, f, sum_tree, min_tree);
    }
    return check_launch();
}

int hexgnn_per_sample(int capacity_pow2, int size, int b, double beta, const double* u, const double* sum_tree,
                      const double* min_tree, int* out_idx, float* out_w, hexgnn_stream_t stream_) {
    if (capacity_pow2 < 1 || (capacity_pow2 & (capacity_pow2 - 1)) || size < 1 || size > capacity_pow2 || b < 1 || !u ||
        !sum_tree || !min_tree || !out_idx || !out_w)
        return HEXGNN_EINVAL;
    per_sample_kernel<<<(b + 255) / 256, 256, 0, (hipStream_t)stream_>>>(capacity_pow2, size, b, beta, u, sum_tree,
                                                                         min_tree, out_idx, out_w);
    return check_launch();
}

}  // extern "C"
