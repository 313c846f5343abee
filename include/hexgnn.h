/* hexgnn.h -- C ABI of libhexgnn.so: the MI355X (gfx950) hot path of GNN_Hex's RainbowDQN loop.
 *
 * The reference (yannikkellerde/GNN_Hex) has no FFI for this path: its boundary is two Python
 * surfaces, GN0.models.get_pre_defined("modern_two_headed") and
 * graph_game.multi_env_manager.Env_manager.  gnn_hex_amd/ mirrors those surfaces in Python and
 * binds THIS header through ctypes; every entry point below names the reference code it replaces.
 *
 * Conventions
 *   - plain pointers and sizes; every pointer is a DEVICE pointer unless its comment says HOST.
 *   - stream-ordered: work is enqueued on `stream` (a hipStream_t passed as void*); no call
 *     synchronises, allocates or frees device memory (scratch is passed in, sized by *_bytes()).
 *   - returns 0 (HEXGNN_OK) or a negative HEXGNN_E* code; never throws.
 *   - node features are fp32 rows padded to HP = hexgnn_padded_width(hidden) = 16*ceil(hidden/16)
 *     floats ("padded layout"); pad columns are zero on output and must be zero on input.
 *   - indices on the device are int32 (N, E < 2^31); int64 only where the reference API has int64.
 *   - re-entrant; a hexgnn_env handle must not be used from two threads at once.
 */
#ifndef HEXGNN_H
#define HEXGNN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HEXGNN_ABI_VERSION 6

#define HEXGNN_OK 0
#define HEXGNN_EINVAL (-1)       /* bad argument (null pointer, negative size, ...) */
#define HEXGNN_EUNSUPPORTED (-2) /* shape outside the compiled kernels (hidden > 256 -- > 128 for the norm / two_headed / HexAra
                                    entry points --, c_in > 8, ...) */
#define HEXGNN_EWORKSPACE (-3)   /* workspace smaller than the matching *_bytes() query */
#define HEXGNN_EHIP (-4)         /* HIP runtime reported an error at launch (see hexgnn_last_hip_error) */
#define HEXGNN_ETIMEOUT (-5)     /* a one-launch stack kernel gave up waiting in its grid barrier (an EARLIER call; sticky once) */

typedef void* hexgnn_stream_t; /* hipStream_t */

/* ---- library ------------------------------------------------------------------------------ */
int hexgnn_abi_version(void);
const char* hexgnn_strerror(int code);
int hexgnn_last_hip_error(void);      /* last hipError_t seen by this thread's failing call */
int hexgnn_padded_width(int hidden);  /* HP; <0 if unsupported */

/* ---- graph structure ---------------------------------------------------------------------- */
/* Replaces the per-call edge_index indexing of torch_geometric MessagePassing.propagate
 * (x[edge_index[0]] gather + scatter onto edge_index[1]; call site GN0/models.py:276): the COO
 * list is sorted ONCE per batch into a target-major CSR (rowptr/col: in-neighbours of each node,
 * ascending) and its transpose (rowptr_t/col_t: out-neighbours, used by the backward gather).
 * invdeg[i] = 1/max(in_degree(i),1) (torch_scatter mean: count.clamp(min=1)).
 * status[0] |= 1 if an index was outside [0,n) (such edges are dropped).  When rowptr, rowptr_t, status and
 * workspace are laid out contiguously in that order the build zeroes them with one memset instead of four.
 * src/dst: the two rows of edge_index (int64, length e). */
size_t hexgnn_csr_workspace_bytes(int n, int e);
int hexgnn_csr_build(int n, int e, const int64_t* src, const int64_t* dst,
                     int* rowptr /*[n+1]*/, int* col /*[e]*/, int* rowptr_t /*[n+1]*/, int* col_t /*[e]*/,
                     float* invdeg /*[n]*/, int* status /*[1]*/,
                     void* workspace, size_t workspace_bytes, hexgnn_stream_t stream);

/* The same result in ONE launch (one workgroup per graph) for batches whose edge list is grouped by graph in graph
 * order -- graph(dst[i]) non-decreasing, what Batch.from_data_list / collate_batch (cpp_hex/hex_graph_game/util.cpp:22-41)
 * and hexgnn_env_observe produce.  Node ranges of the graphs: gptr (int32 [b+1]) or, when ptr64 is given, the int64 `ptr` of
 * the batch, in which case gptr_out [b+1] receives the int32 copy the other calls take.  status bits as above plus
 * 8 = a graph has more than 2048 nodes (use hexgnn_csr_build).  Unlike hexgnn_csr_build this call does NOT clear the
 * status word (caller-zeroed, OR-ed into, like the fused calls): one long-lived "sticky" error word per device serves
 * every batch without a memset launch per step. */
int hexgnn_csr_build_grouped(int n, int e, int b, const int64_t* src, const int64_t* dst, const int* gptr,
                             const int64_t* ptr64, int* gptr_out, int* rowptr, int* col, int* rowptr_t, int* col_t,
                             float* invdeg, int* status, hexgnn_stream_t stream);

/* hexgnn_csr_build_grouped + the weight pack of the network call that follows (hexgnn_qnet_forward /
 * hexgnn_sage_stack_forward over the same c_in / hidden / num_layers and the same wpack) in ONE launch: the two do not depend on
 * each other.  The forward call is then given wl = bl = wr = NULL ("packed already").  b >= 1; exact fp32 math only. */
int hexgnn_csr_build_grouped_pack(int n, int e, int b, const int64_t* src, const int64_t* dst, const int* gptr,
                                  const int64_t* ptr64, int* gptr_out, int* rowptr, int* col, int* rowptr_t, int* col_t,
                                  float* invdeg, int* status, int c_in, int hidden, int num_layers,
                                  const float* const* wl, const float* const* bl, const float* const* wr, void* wpack,
                                  hexgnn_stream_t stream);
/* The same with the collation's own edge offsets: edge_ptr64 [b+1] (device, int64; graph g's edges are src/dst[edge_ptr64[g] ..
 * edge_ptr64[g+1]) -- what Batch.from_data_list (torch_geometric 2.2.0's collate, call site GN0/RainbowDQN/evaluate_elo.py:229) knows
 * when it concatenates the graphs' edge_index).  Spares the three dependent search rounds that locate a graph's edge range; the
 * ranges must tile [0, e) in order (status |= 4 otherwise).  NULL: as hexgnn_csr_build_grouped_pack. */
int hexgnn_csr_build_grouped_pack_e(int n, int e, int b, const int64_t* src, const int64_t* dst, const int* gptr,
                                    const int64_t* ptr64, const int64_t* edge_ptr64, int* gptr_out, int* rowptr, int* col,
                                    int* rowptr_t, int* col_t, float* invdeg, int* status, int c_in, int hidden,
                                    int num_layers, const float* const* wl, const float* const* bl,
                                    const float* const* wr, void* wpack, hexgnn_stream_t stream);
/* + a row-block table for the one-launch stack kernels built ON THE DEVICE in the same launch, for callers whose host never saw
 * the graph sizes (raw tensors of another collation -- the reference's loop hands the model torch_geometric's Batch,
 * GN0/models.py:537): block_starts_out [max_blocks + 1] receives the partition gnn_hex_amd.data.blocks_for_order would give for
 * this graph order, unused entries = n (empty blocks: such a workgroup exits at once), or the plain 128-row partition when the
 * aligned one needs more than max_blocks blocks.  Pass it on as (block_starts, num_blocks = max_blocks) of
 * hexgnn_sage_stack_*_blocks.  Requires ceil(n / 128) <= max_blocks <= 512. */
int hexgnn_csr_build_grouped_pack_b(int n, int e, int b, const int64_t* src, const int64_t* dst, const int* gptr,
                                    const int64_t* ptr64, const int64_t* edge_ptr64, int* gptr_out, int* rowptr, int* col,
                                    int* rowptr_t, int* col_t, float* invdeg, int* status, int c_in, int hidden,
                                    int num_layers, const float* const* wl, const float* const* bl,
                                    const float* const* wr, void* wpack, int* block_starts_out, int max_blocks,
                                    hexgnn_stream_t stream);

/* Replaces the segment lookup inside torch_scatter.scatter(x, graph_indices) (GN0/models.py:381,578)
 * and torch_geometric Batch.ptr: batch (int64, sorted ascending, values in [0,b)) -> gptr[b+1]. */
int hexgnn_graph_ptr(int n, int b, const int64_t* batch, int* gptr /*[b+1]*/, hexgnn_stream_t stream);

/* ---- GraphSAGE stack (CachifiedGNN.forward, GN0/models.py:261-294; SAGEConv semantics
 *      GN0/torch_script_models.py:52-73): y_i = W_l * mean_{j in N(i)} x_j + b_l + W_r * x_i, ReLU
 *      after every layer.  Layer 0 maps c_in -> hidden, the rest hidden -> hidden.
 *      c_in <= 8 (raw features, row stride x_stride floats) or c_in == hidden (padded layout).
 *      hidden <= 128: LDS-resident kernels; hidden 129..256 (grow_width, GN0/models.py:187-238): plain kernels behind the
 *      same calls -- `saved` is then REQUIRED by the forward call whatever need_backward says (every layer's aggregate is
 *      materialised there).
 *      wl/bl/wr: HOST arrays (num_layers entries) of device pointers to the torch parameters
 *      lin_l.weight [out,in], lin_l.bias [out], lin_r.weight [out,in].
 *      flags: HEXGNN_SAGE_LINEAR_LAST = no ReLU after the LAST layer of the stack; a 1-layer stack with it is a bare
 *      SAGEConv.forward (torch_geometric SAGEConv as restated at GN0/torch_script_models.py:52-73). */
#define HEXGNN_SAGE_LINEAR_LAST 1
/* backward only: dy already IS G_{L-1} = dy * [y_{L-1} > 0], written by its producer at
 * (float*)workspace + (size_t)(num_layers - 1) * n * HP (hexgnn_head_backward with HEXGNN_HEAD_MASK_DH does): the stack's
 * first step, a masked copy of dy into that slab, is skipped. */
#define HEXGNN_SAGE_DY_IN_PLACE 2
size_t hexgnn_sage_stack_pack_bytes(int c_in, int hidden, int num_layers);
size_t hexgnn_sage_stack_saved_bytes(int n, int c_in, int hidden, int num_layers);
/* acts:  [num_layers][n][HP] outputs of every layer (post-ReLU); the last slab is the result.
 * saved: aggregated inputs of every layer, kept for the backward pass (may be NULL when
 *        need_backward == 0).  wpack: scratch of hexgnn_sage_stack_pack_bytes(), also read by the
 *        backward call of the same step. */
int hexgnn_sage_stack_forward(int n, int c_in, int hidden, int num_layers,
                              const int* rowptr, const int* col, const float* invdeg,
                              const float* x, int x_stride,
                              const float* const* wl, const float* const* bl, const float* const* wr,
                              void* wpack, float* acts, void* saved, int need_backward, int flags,
                              hexgnn_stream_t stream);

size_t hexgnn_sage_stack_backward_workspace_bytes(int n, int c_in, int hidden, int num_layers);
/* dy: gradient w.r.t. the stack output, padded layout [n][HP].
 * dx: gradient w.r.t. the stack input, padded layout [n][HP]; NULL to skip (always skipped when
 *     c_in != hidden).  d_wl/d_bl/d_wr: HOST arrays of device pointers, written (not accumulated),
 *     same shapes as the parameters.  Deterministic: fixed reduction order, no float atomics. */
int hexgnn_sage_stack_backward(int n, int c_in, int hidden, int num_layers,
                               const int* rowptr, const int* col,
                               const int* rowptr_t, const int* col_t, const float* invdeg,
                               const float* x, int x_stride,
                               const float* acts, const void* saved, const void* wpack,
                               const float* dy, float* dx,
                               float* const* d_wl, float* const* d_bl, float* const* d_wr,
                               void* workspace, size_t workspace_bytes, int flags /* as in the forward call */,
                               hexgnn_stream_t stream);

/* The same call with a TAP: tap_out [n][HP] (may be NULL) receives the gradient w.r.t. the OUTPUT of layer tap_layer
 * (0 <= tap_layer < num_layers - 1) -- DuellingTwoHeaded's final_conv_grads when body and head layers run as one stack
 * (GN0/models.py:553-554: embeds.register_hook). */
int hexgnn_sage_stack_backward_tap(int n, int c_in, int hidden, int num_layers,
                                   const int* rowptr, const int* col,
                                   const int* rowptr_t, const int* col_t, const float* invdeg,
                                   const float* x, int x_stride,
                                   const float* acts, const void* saved, const void* wpack,
                                   const float* dy, float* dx,
                                   float* const* d_wl, float* const* d_bl, float* const* d_wr,
                                   void* workspace, size_t workspace_bytes, int flags,
                                   int tap_layer, float* tap_out, hexgnn_stream_t stream);

/* Both directions with GRAPH-ALIGNED row blocks (round 4).  The one-launch stack kernels give every workgroup a block of at
 * most 128 consecutive rows; by default block b = rows [128 b, 128 b + 128), which cuts graphs wherever a multiple of 128 falls,
 * and a block that holds part of a graph waits, layer by layer, for the blocks that hold the rest.  block_starts (DEVICE array,
 * num_blocks + 1 ascending row offsets, block_starts[0] = 0, block_starts[num_blocks] = n, pieces of at most 128 rows) lets the
 * collation decide instead: with the graphs ordered so that blocks hold whole graphs (gnn_hex_amd.data.pack_order; the
 * reference's Batch.from_data_list, GN0/RainbowDQN/Rainbow/common/utils.py via torch_geometric, keeps the caller's order, and
 * the order of a replay batch carries no meaning) only graphs above 128 rows couple blocks -- the two or three they span.
 * The table's content is checked by the kernel (a block whose range is not such a piece computes nothing and the launch
 * reports HEXGNN_EINVAL through hexgnn_stack_status); num_blocks must lie in [ceil(n / 128), 512]; empty blocks are allowed (a
 * table padded at its end with entries == n, as hexgnn_csr_build_grouped_pack_b writes it: such a workgroup exits at once).
 * A table with more blocks than hexgnn_stack_block_budget() is ignored (default blocks).  Results equal the default
 * blocks' to fp32 rounding (the order in which a row's neighbour sum takes in-block and out-of-block neighbours differs).
 * block_starts == NULL (num_blocks 0): exactly the calls above.  Per-layer launches (batch too large for one resident
 * workgroup per CU, HEXGNN_NO_PERSIST) and hidden > 128 ignore the table. */
int hexgnn_sage_stack_forward_blocks(int n, int c_in, int hidden, int num_layers, const int* rowptr, const int* col,
                                     const float* invdeg, const float* x, int x_stride, const float* const* wl,
                                     const float* const* bl, const float* const* wr, void* wpack, float* acts,
                                     void* saved, int need_backward, int flags, const int* block_starts, int num_blocks,
                                     hexgnn_stream_t stream);
int hexgnn_sage_stack_backward_blocks(int n, int c_in, int hidden, int num_layers, const int* rowptr, const int* col,
                                      const int* rowptr_t, const int* col_t, const float* invdeg, const float* x,
                                      int x_stride, const float* acts, const void* saved, const void* wpack,
                                      const float* dy, float* dx, float* const* d_wl, float* const* d_bl,
                                      float* const* d_wr, void* workspace, size_t workspace_bytes, int flags,
                                      int tap_layer, float* tap_out, const int* block_starts, int num_blocks,
                                      hexgnn_stream_t stream);

/* ---- head tail: HeadNetwork.forward after its gnn (GN0/models.py:374-384), MLP value head
 *      (GN0/models.py:36-82: Linear(4H,H/2) -> relu -> Linear(H/2,1)) and the dueling combine of
 *      DuellingTwoHeaded.forward (GN0/models.py:567-584).
 *      mode 0: q[n]   = tanh(v)[g] + 2tanh(a) - mean_g(2tanh(a))
 *      mode 1: out_v[b] = tanh(v), q[n] = 2tanh(a) - mean_g(2tanh(a))          (seperate=True)
 *      mode 2: q[n]   = 2tanh(a); pooling / value path skipped                  (advantages_only=True)
 *      mode 3: q[n]   = a (raw advantage linear), out_v[b] = v (raw value MLP): HeadNetwork.forward itself,
 *              GN0/models.py:368-384, before DuellingTwoHeaded's activations
 *      mode 4: q[n]   = a only; pooling / value path skipped     (HeadNetwork.forward(advantages_only=True))
 *      Pooled order is [sum | max | min | mean] (value_aggr_types, GN0/models.py:940); max/min route
 *      their gradient to the FIRST row attaining the extremum (torch_scatter CPU kernel). -------- */
size_t hexgnn_head_saved_bytes(int n, int b, int hidden);
int hexgnn_head_forward(int n, int b, int hidden, int mode, const int* gptr, const float* h /*[n][HP]*/,
                        const float* lin_w /*[hidden]*/, const float* lin_b /*[1]*/,
                        const float* v0_w /*[hidden/2][4*hidden]*/, const float* v0_b /*[hidden/2]*/,
                        const float* v1_w /*[hidden/2]*/, const float* v1_b /*[1]*/,
                        float* q /*[n]*/, float* out_v /*[b] (modes 1, 3) or NULL*/,
                        void* saved, hexgnn_stream_t stream);
size_t hexgnn_head_backward_workspace_bytes(int n, int b, int hidden);
/* mode | HEXGNN_HEAD_MASK_DH (backward only): dh is written masked by [h > 0] -- h is the output of a ReLU layer, and the
 * gradient handed to that layer's backward is dh * [h > 0] anyway (saves the stack's masked copy, HEXGNN_SAGE_DY_IN_PLACE). */
#define HEXGNN_HEAD_MASK_DH 8
/* dq: gradient of q [n]; d_out_v: gradient of out_v [b] (modes 1, 3) or NULL.
 * dh: [n][HP] gradient w.r.t. h (padded layout, written).  Parameter gradients written. */
int hexgnn_head_backward(int n, int b, int hidden, int mode, const int* gptr, const float* h,
                         const float* lin_w, const float* v0_w, const float* v1_w,
                         const void* saved, const float* dq, const float* d_out_v,
                         float* dh, float* d_lin_w, float* d_lin_b,
                         float* d_v0_w, float* d_v0_b, float* d_v1_w, float* d_v1_b,
                         void* workspace, size_t workspace_bytes, hexgnn_stream_t stream);

/* ---- head tail of the `two_headed` family (get_pre_defined("two_headed"), GN0/models.py:901-918): HeadNetwork with
 *      value_head_type="linear" over value_aggr_types=("mean",) (GN0/models.py:319-330,374-384): value_g = val_w . mean_{i in g} h_i
 *      + val_b; advantage linear, dueling combine and modes 0..4 exactly as hexgnn_head_forward / _backward above. ---- */
size_t hexgnn_head_linear_saved_bytes(int n, int b);
int hexgnn_head_linear_forward(int n, int b, int hidden, int mode, const int* gptr, const float* h /*[n][HP]*/,
                               const float* lin_w /*[hidden]*/, const float* lin_b /*[1]*/,
                               const float* val_w /*[hidden]*/, const float* val_b /*[1]*/,
                               float* q /*[n]*/, float* out_v /*[b] (modes 1, 3) or NULL*/,
                               void* saved, hexgnn_stream_t stream);
size_t hexgnn_head_linear_backward_workspace_bytes(int n, int b, int hidden);
int hexgnn_head_linear_backward(int n, int b, int hidden, int mode /* | HEXGNN_HEAD_MASK_DH */, const int* gptr,
                                const float* h, const float* lin_w, const float* val_w, const void* saved,
                                const float* dq, const float* d_out_v, float* dh,
                                float* d_lin_w, float* d_lin_b, float* d_val_w, float* d_val_b,
                                void* workspace, size_t workspace_bytes, hexgnn_stream_t stream);

/* ---- HexAra policy/value network SAGE_torch_script (GN0/torch_script_models.py:286-379), the pieces the SAGE stack and
 *      the head tail above do not cover:
 *      (a) the policy head's last layer SAGEConv(H, 1) (ModifiedBaseNet with out_channels=1, lines 123-144,296):
 *          out[i] = bias + wr . h_i + mean_{j in N(i)} wl . h_j.   dots: [n][2] scratch (kept for nothing: recomputed backward).
 *      (b) lines 326-378: terminal nodes (the first two rows of every graph) dropped, the graph's swap logit appended to its
 *          segment when swapping is allowed there (feature 2 of the graph's last row; of its FIRST row for the last graph),
 *          scatter_log_softmax per segment.  out_pi / out_gi: capacity n - 2b + b entries; out_ptr [b+1] =
 *          output_batch_ptr (out_ptr[b] = number of entries written).  Pinned by rl_loop/unittest_model.py:16-92. ---- */
int hexgnn_sage_scalar_forward(int n, int hidden, const int* rowptr, const int* col, const float* invdeg,
                               const float* h /*[n][HP]*/, const float* wl /*[hidden]*/, const float* wr /*[hidden]*/,
                               const float* bias /*[1]*/, float* out /*[n]*/, float* dots /*[n][2]*/,
                               hexgnn_stream_t stream);
size_t hexgnn_sage_scalar_backward_workspace_bytes(int n, int hidden);
int hexgnn_sage_scalar_backward(int n, int hidden, const int* rowptr_t, const int* col_t, const float* invdeg,
                                const float* h, const float* wl, const float* wr, const float* dout /*[n]*/,
                                float* dh /*[n][HP]*/, float* d_wl, float* d_wr, float* d_bias,
                                void* workspace, size_t workspace_bytes, hexgnn_stream_t stream);
int hexgnn_policy_log_softmax_forward(int n, int b, const int* gptr, const float* x /*[n][x_stride], feature 2 read*/,
                                      int x_stride, int swap_allowed, const float* pi_raw /*[n]*/,
                                      const float* should_swap /*[b] or NULL*/, float* out_pi, int64_t* out_gi,
                                      int64_t* out_ptr /*[b+1]*/, hexgnn_stream_t stream);
int hexgnn_policy_log_softmax_backward(int n, int b, const int* gptr, const float* x, int x_stride, int swap_allowed,
                                       const int64_t* out_ptr, const float* out_pi, const float* d_out,
                                       float* d_pi_raw /*[n]*/, float* d_should_swap /*[b] or NULL*/,
                                       hexgnn_stream_t stream);

/* ---- whole-batch LayerNorm (--norm=True): torch_geometric 2.2.0 LayerNorm(hidden, mode="graph") as
 *      CachifiedGNN.forward / DuellingTwoHeaded.forward call it, WITHOUT a batch vector (GN0/models.py:8,286-287,550-551,
 *      935,945): mean and biased std over ALL n x hidden elements, y = (x - mean) / (std + eps) * weight + bias, then the
 *      activation CachifiedGNN applies after the norm (relu != 0).  x / y / dy / dx: padded layout [n][HP]; stats [2]
 *      receives (mean, 1/(std+eps)) for the backward call.  Deterministic (fp64 block partials, fixed order). ---- */
size_t hexgnn_graph_layernorm_workspace_bytes(int hidden);
int hexgnn_graph_layernorm_forward(int n, int hidden, const float* x, const float* weight /*[hidden]*/,
                                   const float* bias /*[hidden]*/, float eps, int relu, float* y, float* stats /*[2]*/,
                                   void* workspace, size_t workspace_bytes, hexgnn_stream_t stream);
/* The same over a CAPACITY-sized buffer whose batch is its first *n_live rows (n_live: device int, clamped to [0, n]; NULL =
 * all n rows): the closed acting loop (hexgnn_env_observe -> forward -> hexgnn_select_actions -> hexgnn_env_step, SURVEY 8(a)
 * envs; the reference rebuilds an exact-size batch per move, graph_game/multi_env_manager.py:76-103) keeps its node count on
 * the device.  Identical, bit for bit, to the call above with n = *n_live; rows at and after *n_live are left untouched. */
int hexgnn_graph_layernorm_forward_live(int n, const int* n_live, int hidden, const float* x, const float* weight,
                                        const float* bias, float eps, int relu, float* y, float* stats, void* workspace,
                                        size_t workspace_bytes, hexgnn_stream_t stream);
/* y: the forward output (only read when relu != 0, for the mask); d_weight / d_bias [hidden] are written. */
int hexgnn_graph_layernorm_backward(int n, int hidden, const float* x, const float* y, const float* weight,
                                    const float* stats, const float* dy, float eps, int relu, float* dx,
                                    float* d_weight, float* d_bias, void* workspace, size_t workspace_bytes,
                                    hexgnn_stream_t stream);

/* ---- CachedGraphNorm (GN0/models.py:644-670; torch_geometric GraphNorm + a statistics cache) as CachifiedGNN.forward calls
 *      it, WITHOUT a batch vector (GN0/models.py:282-283): per-CHANNEL statistics over all n nodes of the batch,
 *          mean_c = mean_i x_ic;  o = x - mean * mean_scale;  var_c = mean_i o_ic^2;  y = weight * o / sqrt(var + eps) + bias,
 *      then the activation CachifiedGNN applies after the norm (relu != 0).  stats [2][HP] = mean | var: WRITTEN when
 *      use_cache == 0 (the caller keeps them as mean_cache / var_cache on set_cache), READ when use_cache != 0 (eval mode after
 *      a set_cache forward: the statistics are constants, also in the backward).  Deterministic (fp64 column partials). ---- */
size_t hexgnn_graph_colnorm_workspace_bytes(int hidden);
int hexgnn_graph_colnorm_forward(int n, int hidden, const float* x, const float* weight /*[hidden]*/,
                                 const float* bias /*[hidden]*/, const float* mean_scale /*[hidden]*/, float eps, int relu,
                                 int use_cache, float* y, float* stats /*[2][HP]*/,
                                 void* workspace, size_t workspace_bytes, hexgnn_stream_t stream);
int hexgnn_graph_colnorm_backward(int n, int hidden, const float* x, const float* y, const float* weight,
                                  const float* mean_scale, const float* stats, const float* dy, float eps, int relu,
                                  int use_cache, float* dx, float* d_weight, float* d_bias, float* d_mean_scale,
                                  void* workspace, size_t workspace_bytes, hexgnn_stream_t stream);

/* ---- SAGE stack with that LayerNorm between every layer's contraction and its ReLU (CachifiedGNN.forward with norms,
 *      GN0/models.py:261-294: x = relu(norm_l(conv_l(x)))): the per-layer sequence above in one call per direction -- weights
 *      packed once, ALL layers' weight gradients in one batched GEMM.  nw / nb / d_nw / d_nb: HOST arrays of device pointers
 *      to the norms' weight / bias [hidden] (and their gradients).  pre: [L][n][HP] contraction outputs, acts: [L][n][HP]
 *      norm + ReLU outputs (the last slab is the result), stats: [L][2]; all three plus saved / wpack go from the forward to
 *      the backward call.  norm_ws: hexgnn_graph_layernorm_workspace_bytes(hidden). */
int hexgnn_sage_norm_stack_forward(int n, int c_in, int hidden, int num_layers, const int* rowptr, const int* col,
                                   const float* invdeg, const float* x, int x_stride, const float* const* wl,
                                   const float* const* bl, const float* const* wr, const float* const* nw,
                                   const float* const* nb, float eps, void* wpack, float* pre, float* acts, void* saved,
                                   float* stats, void* norm_ws, size_t norm_ws_bytes, int need_backward,
                                   hexgnn_stream_t stream);
/* Forward-only (need_backward must be 0 when n_live is given) over capacity-sized buffers: the SAGE layers run over all n rows
 * (rowptr must describe empty rows behind the live ones), every norm's statistics cover the first *n_live rows only. */
int hexgnn_sage_norm_stack_forward_live(int n, const int* n_live, int c_in, int hidden, int num_layers, const int* rowptr,
                                        const int* col, const float* invdeg, const float* x, int x_stride,
                                        const float* const* wl, const float* const* bl, const float* const* wr,
                                        const float* const* nw, const float* const* nb, float eps, void* wpack, float* pre,
                                        float* acts, void* saved, float* stats, void* norm_ws, size_t norm_ws_bytes,
                                        int need_backward, hexgnn_stream_t stream);
size_t hexgnn_sage_norm_stack_backward_workspace_bytes(int n, int c_in, int hidden, int num_layers);
int hexgnn_sage_norm_stack_backward(int n, int c_in, int hidden, int num_layers, const int* rowptr_t, const int* col_t,
                                    const float* invdeg, const float* x, int x_stride, const float* pre,
                                    const float* acts, const void* saved, const void* wpack, const float* stats,
                                    const float* const* nw, float eps, const float* dy, float* dx, float* const* d_wl,
                                    float* const* d_bl, float* const* d_wr, float* const* d_nw, float* const* d_nb,
                                    void* workspace, size_t workspace_bytes, void* norm_ws, size_t norm_ws_bytes,
                                    hexgnn_stream_t stream);

/* ---- fused per-graph path: the WHOLE network (raw first layer, all body + head SAGE layers, head tail) in one
 *      launch per direction, one workgroup per graph with the node features resident in LDS.  Usable when
 *      hexgnn_qnet_supported(): hidden <= 112, c_in <= 8, every graph <= 128 nodes (status |= 2 and the graph is
 *      skipped otherwise; status |= 4: an edge leaves its graph).  status is OR-ed into, never cleared, so the
 *      word written by hexgnn_csr_build (|= 1: node id out of range) can be shared.  wl/bl/wr list the body layers followed by the
 *      layers of the head that is evaluated (HOST arrays, total_layers entries).  saved/wpack/acts are produced
 *      by the forward call and consumed by the backward call of the same step.  Same results as the
 *      sage_stack + head calls above.
 *      math: 0 = exact fp32 MFMA (v_mfma_f32_16x16x4_f32; same fmaf chains as the layered kernels);
 *            1 = split precision "f16x3": weights (per layer) and rows (per row) are scaled by exact powers of two, every
 *                scaled fp32 operand a = hi + lo (fp16 pair, 22 significand bits), W*x ~= Whi*xhi + Whi*xlo + Wlo*xhi on
 *                the f16 MFMA pipe with fp32 accumulation, scales undone exactly in the epilogue (product error
 *                ~3*2^-22 of max|w| max|x|; same parity bar as math 0, see tests/test_gpu_model.py).  The backward
 *                call must use the math of its forward. ---- */
int hexgnn_qnet_supported(int c_in, int hidden, int max_nodes_per_graph);
size_t hexgnn_qnet_saved_bytes(int n, int b, int c_in, int hidden, int total_layers);
int hexgnn_qnet_forward(int n, int b, int c_in, int hidden, int total_layers, int mode, const int* gptr,
                        const int* rowptr, const int* col, const float* invdeg, const float* x, int x_stride,
                        const float* const* wl, const float* const* bl, const float* const* wr,
                        const float* lin_w, const float* lin_b, const float* v0_w, const float* v0_b,
                        const float* v1_w, const float* v1_b,
                        void* wpack /* hexgnn_sage_stack_pack_bytes(c_in, hidden, total_layers) */,
                        float* acts /*[total_layers][n][HP]*/, void* saved, int need_backward,
                        int acts_layer /* need_backward == 0: store only this layer's activations (the body output read as
                                          final_conv_acts); -1: every layer's */,
                        int math,
                        float* q /*[n]*/, float* out_v /*[b] (mode 1) or NULL*/, int* status /*[1], caller-zeroed*/,
                        hexgnn_stream_t stream);
size_t hexgnn_qnet_backward_workspace_bytes(int n, int b, int c_in, int hidden, int total_layers);
/* d_embeds: optional [n][HP] gradient w.r.t. the output of body layer body_layers-1 (final_conv_grads). */
int hexgnn_qnet_backward(int n, int b, int c_in, int hidden, int total_layers, int body_layers, int mode, int math,
                         const int* gptr, const int* rowptr_t, const int* col_t, const float* invdeg,
                         const float* x, int x_stride, const float* acts, const void* saved, const void* wpack,
                         const float* lin_w, const float* v0_w, const float* v1_w,
                         const float* dq, const float* d_out_v, float* d_embeds,
                         float* const* d_wl, float* const* d_bl, float* const* d_wr,
                         float* d_lin_w, float* d_lin_b, float* d_v0_w, float* d_v0_b, float* d_v1_w, float* d_v1_b,
                         void* workspace, size_t workspace_bytes, int* status, hexgnn_stream_t stream);

/* The same backward in stages, so that the gradient all-reduce of the layers that are finished can run (RCCL, its own
 * stream) while the weight-gradient GEMM of the remaining layers still computes (the reference has no
 * collective, README.md:53).  stages: bit set of
 *   HEXGNN_QBWD_DATA    the data chain (G of every layer, per-graph partials); must come first
 *   HEXGNN_QBWD_SMALL   per-graph partials -> gradients of the raw first layer, the advantage linear and the value MLP
 *   HEXGNN_QBWD_HIDDEN  weight-gradient GEMM + slice reduce of the hidden-input layers [layer_lo, layer_hi), 1 <= lo <= hi <= total_layers
 * Every call takes the full argument list of hexgnn_qnet_backward (same workspace); the stages of one step may be issued in
 * any number of calls, each layer range once.  hexgnn_qnet_backward == all three stages over [1, total_layers). */
#define HEXGNN_QBWD_DATA 1
#define HEXGNN_QBWD_SMALL 2
#define HEXGNN_QBWD_HIDDEN 4
int hexgnn_qnet_backward_staged(int n, int b, int c_in, int hidden, int total_layers, int body_layers, int mode, int math,
                                const int* gptr, const int* rowptr_t, const int* col_t, const float* invdeg,
                                const float* x, int x_stride, const float* acts, const void* saved, const void* wpack,
                                const float* lin_w, const float* v0_w, const float* v1_w,
                                const float* dq, const float* d_out_v, float* d_embeds,
                                float* const* d_wl, float* const* d_bl, float* const* d_wr,
                                float* d_lin_w, float* d_lin_b, float* d_v0_w, float* d_v0_b, float* d_v1_w, float* d_v1_b,
                                void* workspace, size_t workspace_bytes, int* status,
                                int stages, int layer_lo, int layer_hi, hexgnn_stream_t stream);

/* The staged call with ALL parameter gradients in ONE flat fp32 buffer (the layout the single RCCL all-reduce uses):
 * gradient k lives at flat + offsets[k], offsets = HOST array of 3 * total_layers + 6 element offsets in the order
 * (d_wl[l], d_bl[l], d_wr[l]) for l = 0 .. total_layers-1, then d_lin_w, d_lin_b, d_v0_w, d_v0_b, d_v1_w, d_v1_b.  One base
 * pointer + a table the host caches per model instead of 3 * total_layers + 6 pointers rebuilt per step (the eager path
 * of the Python mirror: GN0/models.py:537-584 driven by `loss.backward()`). */
int hexgnn_qnet_backward_flat(int n, int b, int c_in, int hidden, int total_layers, int body_layers, int mode, int math,
                              const int* gptr, const int* rowptr_t, const int* col_t, const float* invdeg,
                              const float* x, int x_stride, const float* acts, const void* saved, const void* wpack,
                              const float* lin_w, const float* v0_w, const float* v1_w,
                              const float* dq, const float* d_out_v, float* d_embeds,
                              float* flat, const int64_t* offsets /* HOST */,
                              void* workspace, size_t workspace_bytes, int* status,
                              int stages, int layer_lo, int layer_hi, hexgnn_stream_t stream);

/* One-launch SAGE stack kernels behind hexgnn_sage_stack_* (all hidden layers of a stack in one launch when every workgroup can be
 * resident; GN0/models.py:261-294's layer loop).  Their waits have a poll budget: a launch that exhausts it writes NaN rows
 * into its own output and sets HEXGNN_ETIMEOUT in a status word that the NEXT stack call returns -- or this query, at any point
 * where the caller has synchronised (clear != 0 resets it).  hexgnn_stack_reserve_cus: CUs the residency guard leaves to kernels
 * that run BESIDE a stack kernel (RCCL channels while the gradient all-reduce overlaps the backward); returns the previous
 * value.  The guard also keeps to per-layer launches while a one-launch kernel of this process is in flight on another stream. */
int hexgnn_stack_status(int clear);
int hexgnn_stack_reserve_cus(int cus);
/* Row blocks a one-launch stack kernel may have right now on the current device (one resident workgroup per CU, minus the
 * reserved CUs; 0 under a CU mask): the budget a collation packs against (gnn_hex_amd.data.pack_order's max_blocks). */
int hexgnn_stack_block_budget(void);
/* Test aids (tests/test_gpu_stack_stress.py): persist -1 = default (HEXGNN_NO_PERSIST decides), 0 = per-layer launches, 1 = one
 * launch where the guard allows; skew_seed != 0 delays every workgroup by a pseudo-random time per layer.  hexgnn_debug_occupy:
 * `blocks` 1024-thread workgroups streaming `buffer` for ~usec microseconds on `stream` (uneven load beside a stack kernel). */
int hexgnn_debug_stack_mode(int persist, unsigned skew_seed);
int hexgnn_debug_occupy(int blocks, int usec, const void* buffer, size_t buffer_bytes, void* sink /* >= 16 B */,
                        hexgnn_stream_t stream);

/* The DQN update's loss folded into the network calls (ABI 6): `loss = loss_fn(Q[sel], target)` of the RainbowDQN training step
 * (README.md:5,7: --loss_fn=mse, --prioritized_er=True importance weights; the step of SURVEY.md 8d) is graph-local when the
 * update selects ONE node per graph, sel[g] a row of graph g (the action taken in the sampled transition), so the forward
 * kernel's tail forms it for its own graph: td[g] = Q[sel[g]] - target[g], loss_part[g] = weights[g] * l(td[g]) and
 * dq = d loss / d Q ([n], zero except at the selected rows) -- no hexgnn_td_loss_forward_backward launch between the two
 * network launches.  Same per-entry expressions as that launch (bit-identical td and dq).  A sel[g] outside graph g sets
 * status |= 16 and poisons td[g] / loss_part[g] with NaN.  mode 0, need_backward 1; loss_fn 0 = mse, 1 = Huber(1).
 * hexgnn_qnet_backward_flat_td is hexgnn_qnet_backward_flat (mode 0) whose reduce launch also writes
 * loss[0] = sum(loss_part) / b in hexgnn_td_loss_forward's reduction shape (its bits), when stages has HEXGNN_QBWD_SMALL. */
int hexgnn_qnet_forward_td(int n, int b, int c_in, int hidden, int total_layers, const int* gptr,
                           const int* rowptr, const int* col, const float* invdeg, const float* x, int x_stride,
                           const float* const* wl, const float* const* bl, const float* const* wr,
                           const float* lin_w, const float* lin_b, const float* v0_w, const float* v0_b,
                           const float* v1_w, const float* v1_b, void* wpack, float* acts, void* saved,
                           int math, float* q /*[n]*/, int* status,
                           const int64_t* sel /*[b]*/, const float* target /*[b]*/, const float* weights /*[b] or NULL*/,
                           int loss_fn, float* dq /*[n]*/, float* td /*[b]*/, float* loss_part /*[b]*/,
                           hexgnn_stream_t stream);
int hexgnn_qnet_backward_flat_td(int n, int b, int c_in, int hidden, int total_layers, int body_layers, int math,
                                 const int* gptr, const int* rowptr_t, const int* col_t, const float* invdeg,
                                 const float* x, int x_stride, const float* acts, const void* saved, const void* wpack,
                                 const float* lin_w, const float* v0_w, const float* v1_w,
                                 const float* dq, float* d_embeds, float* flat, const int64_t* offsets /* HOST */,
                                 void* workspace, size_t workspace_bytes, int* status,
                                 int stages, int layer_lo, int layer_hi,
                                 const float* loss_part /*[b]*/, float* loss /*[1]*/, hexgnn_stream_t stream);

/* ---- batched board-graph builder: num_envs lock-stepped Hex / Shannon node-switching games on the device.
 *      Replaces Hex_game / Node_switching_game (graph_game/graph_tools_games.py:20-29,
 *      graph_game/shannon_node_switching_game.py:80-205, graph_game/hex_board_game.py:214-233) as driven by
 *      Env_manager (graph_game/multi_env_manager.py:31-111), and convert_node_switching_game(old_style=True)
 *      (GN0/util/convert_graph.py:60-130) + Batch.from_data_list for the observation.  Vertex ids: 0,1 terminals,
 *      i+2 = board cell i.  Iteration order inside dead_and_captured is the canonical ascending order of
 *      oracle/env_ref.c (the reference's own two implementations disagree on it).
 *      create/destroy allocate (not stream-ordered); everything else is stream-ordered and non-allocating. ------ */
typedef struct hexgnn_env hexgnn_env;
int hexgnn_env_create(int num_envs, int hex_size, hexgnn_env** out);   /* hex_size <= 25; all envs at the start position, maker to move */
void hexgnn_env_destroy(hexgnn_env* env);
int hexgnn_env_num_vertices(const hexgnn_env* env);                    /* hex_size^2 + 2 */
int hexgnn_env_words(const hexgnn_env* env);                           /* 64-bit words per adjacency row */
/* Hex_game(size) for the envs with mask[i] != 0 (all when mask == NULL), gp["m"] = maker_turn.
 * sizes (optional, [num_envs][2]): (nodes, directed edges) of the reset envs. */
int hexgnn_env_reset(hexgnn_env* env, const uint8_t* mask, int maker_turn, int* sizes, hexgnn_stream_t stream);
int hexgnn_env_set_maker_turn(hexgnn_env* env, int maker_turn, hexgnn_stream_t stream);
/* make_move(actions[i], remove_dead_and_captured) + who_won for every env (Env_manager.step, multi_env_manager.py:76-103).
 * actions: vertex ids (int32).  result [num_envs][5] = (winner: -1 none / 0 maker / 1 breaker, total_num_moves at that
 * point, nodes, directed edges, error: 1 = illegal action, env untouched).  With auto_reset a finished env is replaced by a
 * fresh start position whose side to move is reset_maker_turn; nodes/edges then describe the fresh graph.  Once a game
 * is decided only winner and move count are defined: the residual graph of a finished game is unspecified (the maker's
 * winning move short-cuts the bookkeeping nobody reads). */
int hexgnn_env_step(hexgnn_env* env, const int* actions, int remove_dead_and_captured, int auto_reset,
                    int reset_maker_turn, int* result, hexgnn_stream_t stream);
/* Batched observation.  node_off/edge_off: [num_envs+1] exclusive prefix sums of the per-env (nodes, directed edges)
 * reported by step/reset; e_total = edge_off[num_envs].
 *   x [N][3] f32 = (degree, is_terminal, maker_to_move)           backmap [N] i64: rank -> vertex id
 *   edge_local  [2][e_total] i64: per graph the E_g/2 edges (s > t, sorted) then their flipped copies, LOCAL ranks
 *   edge_global [2][e_total] i64: the same with the graph's node offset added (== Batch.edge_index)
 *   rowptr [N+1] / col [e_total] i32 + invdeg [N] f32: the sorted CSR hexgnn_csr_build would produce (symmetric
 *   graph: it is its own transpose)                                 batch_vec [N] i64: graph id of every node */
int hexgnn_env_observe(hexgnn_env* env, const int* node_off, const int* edge_off, int64_t e_total, float* x,
                       int64_t* backmap, int64_t* edge_local, int64_t* edge_global, int* rowptr, int* col,
                       float* invdeg, int64_t* batch_vec, hexgnn_stream_t stream);
/* Raw state dump (tests): adj [num_envs][nv][words] u64, alive [num_envs][nv] u8, maker_turn / total_moves [num_envs],
 * response sets [num_envs][nv] i16 (-1 = none; may be NULL). */
/* Overwrite the whole state of every env with arrays in hexgnn_env_export's layout (all six required). */
int hexgnn_env_import(hexgnn_env* env, const uint64_t* adj, const uint8_t* alive, const int* maker_turn,
                      const int* total_moves, const int16_t* resp_maker, const int16_t* resp_breaker, hexgnn_stream_t stream);
/* node_off / edge_off [k+1] (device) = exclusive prefix sums of result[env][2] / result[env][3] of the last
 * hexgnn_env_step: feeds hexgnn_env_observe without reading the sizes back (device-resident rollouts). */
int hexgnn_env_offsets(int k, const int* result, int* node_off, int* edge_off, hexgnn_stream_t stream);
int hexgnn_env_export(hexgnn_env* env, uint64_t* adj, uint8_t* alive, int* maker_turn, int* total_moves,
                      int16_t* resp_maker, int16_t* resp_breaker, hexgnn_stream_t stream);

/* Same observation, built from ANY array of board states (the replay ring) for the k states listed in `index`
 * (NULL = the first k): adj [num_states][nv][words] u64, alive [num_states][nv] u8, side [num_states] u8 (1 = maker
 * to move).  Replaces re-collating stored torch_geometric Data objects with Batch.from_data_list in the (absent)
 * replay buffer's sample(). */
int hexgnn_states_observe(int hex_size, int k, const uint64_t* adj, const uint8_t* alive, const uint8_t* side,
                          const int* index, const int* node_off, const int* edge_off, int64_t e_total, float* x,
                          int64_t* backmap, int64_t* edge_local, int64_t* edge_global, int* rowptr, int* col,
                          float* invdeg, int64_t* batch_vec, hexgnn_stream_t stream);

/* ---- prioritized replay sampler (RainbowDQN --prioritized_er=True, README.md:5,7; the buffer itself is in the
 *      un-vendored submodule GN0/RainbowDQN/Rainbow, .gitmodules:1-4 -- PARITY UNPINNED, see replay.hip).
 *      Trees: fp64 arrays of 2*capacity entries (capacity a power of two), node 1 = root, leaves [capacity, 2*capacity).
 *      hexgnn_per_update writes prio_alpha[i] (= priority^alpha, computed by the caller) to leaf idx[i] of both trees and
 *      rebuilds the ancestors.  hexgnn_per_sample: stratified proportional sampling with the caller's uniforms u[b] in
 *      [0,1): out_idx[i] = prefix-sum search of (i + u[i]) * total / b; out_w[i] = importance weight normalised by the
 *      largest possible weight (size = number of valid leaves). ---------------------------------------------------- */
int hexgnn_per_init(int capacity_pow2, double* sum_tree, double* min_tree, hexgnn_stream_t stream);
int hexgnn_per_update(int capacity_pow2, int k, const int* idx, const double* prio_alpha, double* sum_tree,
                      double* min_tree, hexgnn_stream_t stream);
/* Fused priority update (one launch): td != NULL: p_i = |td[i]| + eps, leaf idx[i] <- p_i^alpha (last occurrence of a slot
 * wins), *max_priority <- max(*max_priority, max_i p_i); td == NULL: every listed leaf <- (*max_priority)^alpha (a block of
 * new transitions stored at the running maximum).  idx: int32 (idx_bits 32) or int64 (64) slots; out-of-range ones are ignored. */
int hexgnn_per_update_td(int capacity_pow2, int k, const void* idx, int idx_bits, const float* td, double alpha, double eps,
                         double* max_priority, double* sum_tree, double* min_tree, hexgnn_stream_t stream);
int hexgnn_per_sample(int capacity_pow2, int size, int b, double beta, const double* u, const double* sum_tree,
                      const double* min_tree, int* out_idx, float* out_w, hexgnn_stream_t stream);

/* ---- in-library kernel timing: HIP events recorded on the launch stream around every launch of ONE
 *      kernel class (bench.py's live roofline measurement; torch.cuda.Event would only see torch's current
 *      stream and whole calls).  Not for use under graph capture. --------------------------------------- */
#define HEXGNN_K_SAGE_FWD 0   /* sage_hidden_fwd_kernel  (gather + MFMA, one launch per hidden-input layer) */
#define HEXGNN_K_SAGE_BWD 1   /* sage_hidden_bwd_kernel  (gradient gather + MFMA) */
#define HEXGNN_K_SAGE_DW 2    /* sage_dw_kernel          (batched weight-gradient GEMM) */
#define HEXGNN_K_HEAD_FWD 3
#define HEXGNN_K_HEAD_BWD 4
#define HEXGNN_K_SAGE_FIRST 5 /* raw-feature first layer forward */
#define HEXGNN_K_COMBINE 6
#define HEXGNN_K_CSR 7        /* the four CSR-build kernels together */
#define HEXGNN_K_QNET_FWD 8   /* fused per-graph forward (whole network, one launch) */
#define HEXGNN_K_QNET_BWD 9   /* fused per-graph backward data chain */
#define HEXGNN_K_COUNT 10
int hexgnn_profile_enable(int kernel_class); /* -1: off.  Clears earlier samples. */
/* Waits for the recorded events; returns the number of launches and their summed duration. */
int hexgnn_profile_read(int* launches, float* total_ms);

/* ---- TD loss on selected nodes (the training loop's `loss_fn(Q[sel], target)` with the prioritized replay's importance
 *      weights; Rainbow agent of the un-vendored submodule, flags README.md:5,7 --loss_fn=mse --prioritized_er=True):
 *      loss = mean_j w_j * l(q[sel_j] - target_j), l = d^2 (loss_fn 0) or Huber with delta 1 (loss_fn 1); weights may be
 *      NULL.  td[j] = q[sel_j] - target_j (priority update).  backward: dq[n] = d loss / d q scaled by *grad_loss
 *      (device scalar).  sel entries outside [0, n) contribute nothing. ---- */
int hexgnn_td_loss_forward(int n, int k, const float* q, const int64_t* sel, const float* target, const float* weights,
                           int loss_fn, float* loss, float* td, hexgnn_stream_t stream);
int hexgnn_td_loss_backward(int n, int k, const int64_t* sel, const float* td, const float* weights, int loss_fn,
                            const float* grad_loss, float* dq, hexgnn_stream_t stream);

/* Both of the above in ONE launch for the usual case that the loss itself is differentiated (grad_loss == 1, what
 * `loss.backward()` of the training loop does): loss, td AND dq[n] = d loss / d q.  Bit-identical to the two calls. */
int hexgnn_td_loss_forward_backward(int n, int k, const float* q, const int64_t* sel, const float* target,
                                    const float* weights, int loss_fn, float* loss, float* td, float* dq,
                                    hexgnn_stream_t stream);

/* ---- acting: epsilon-greedy action per graph straight from the Q / advantage vector (replaces the per-graph python
 *      argmax over action_values[ptr[g]+2 : ptr[g+1]] of GN0/RainbowDQN/evaluate_elo.py:253-266 and the backmap lookup
 *      of Env_manager.validate_actions, graph_game/multi_env_manager.py:62-64).  u: [b][2] uniforms in [0,1) or NULL for
 *      pure greedy.  backmap / action_vertex may be NULL (ranks only: the double-DQN argmax over a sampled batch).
 *      action_vertex feeds hexgnn_env_step without leaving the device. ---------------------------------- */
int hexgnn_select_actions(int b, const int* gptr, const float* q, const int64_t* backmap, float eps, const float* u,
                          int* action_vertex /*[b]*/, int* action_rank /*[b]*/, uint8_t* exploratory /*[b] or NULL*/,
                          hexgnn_stream_t stream);

/* ---- layout helpers (host tensors <-> padded layout) ---------------------------------------- */
/* dst[n][HP] <- src[n][hidden] (row stride src_stride floats), pad columns zeroed; and back. */
int hexgnn_pad_rows(int n, int hidden, const float* src, int src_stride, float* dst, hexgnn_stream_t stream);
int hexgnn_unpad_rows(int n, int hidden, const float* src, float* dst, int dst_stride, hexgnn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* HEXGNN_H */
