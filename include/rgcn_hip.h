/*
 * rgcn_hip.h - C ABI of librgcn_hip.so: the MI355X (gfx950) R-GCN message-passing
 * engine + DistMult scoring head.
 *
 * The reference (arnold117/PrimeKG-RGCN-LinkPrediction) has no FFI of its own: the
 * path is reached through Python, `torch_geometric.nn.RGCNConv` constructed at
 * src/models/rgcn.py:72-85 and called at rgcn.py:123,128, and `LinkPredictor.forward`
 * (rgcn.py:189-213) called at rgcn.py:329.  Every entry point below names the
 * torch-op sequence inside those calls that it replaces.
 *
 * Conventions
 *  - plain C: pointers and sizes only, no torch types.  All data pointers are DEVICE
 *    pointers unless the name ends in _host.  `stream` is a hipStream_t passed as void*.
 *  - every function is asynchronous on `stream` and does no allocation, no host sync and
 *    no host<->device copy (HIP-graph capturable), EXCEPT rgcn_graph_create /
 *    rgcn_graph_destroy, the one-time bucketing of a static graph.
 *  - inputs are borrowed and never written; outputs are caller-allocated.
 *  - return value: RGCN_OK (0) or a negative RGCN_ERR_* code; rgcn_strerror() names it.
 *    Nothing aborts the process.
 *  - feature dims must be multiples of 4 floats (16-byte rows); fp32 throughout
 *    ("within 1e-5 fp32 of PyG RGCNConv", BASELINE.json north_star).
 */
#ifndef RGCN_HIP_H
#define RGCN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RGCN_ABI_VERSION 24

enum {
  RGCN_OK = 0,
  RGCN_ERR_ARG = -1,          /* null pointer, negative size, dim not a multiple of 4 ... */
  RGCN_ERR_RANGE = -2,        /* node id outside [0,N) or relation id outside [0,R) */
  RGCN_ERR_HIP = -3,          /* a HIP runtime call failed */
  RGCN_ERR_UNSUPPORTED = -4,  /* shape outside what the kernels were built for */
  RGCN_ERR_WORKSPACE = -5     /* workspace pointer null or too small */
};

int rgcn_abi_version(void);
const char* rgcn_strerror(int code);

/* ------------------------------------------------------------------------------------
 * Relation bucketing (row A2: `edge_index[:, edge_type == r]` for every r, done once).
 *
 * Builds, on the device, the two CSR-by-relation structures of a static multigraph:
 *   forward    : segment s = dst*R + rel, col[] = src          (mean over in-edges)
 *   transposed : segment s = src*R + rel, col_t[] = dst,
 *                w_t[] = 1 / cnt[dst*R + rel]                  (what autograd of A4 scatters)
 * via a STABLE radix sort of the edge columns, so inside every segment the edges keep the
 * reference's order-preserving column selection (bit-exact integer work).  Also builds the
 * chunked work lists the aggregate kernels walk (segments longer than 64 edges are split
 * and tree-reduced so that degree skew cannot serialise a launch).
 *
 * edge_index: int64[2,E] row-major (row 0 = source j, row 1 = destination i);
 * edge_type : int64[E].  Returns RGCN_ERR_RANGE (and *out = NULL) when any id is out of
 * range - the reference filters those on the host (src/train.py:571-586).
 * This call synchronises `stream` and allocates device memory owned by the handle.
 * ---------------------------------------------------------------------------------- */
typedef struct rgcn_graph rgcn_graph;

int rgcn_graph_create(const int64_t* edge_index, const int64_t* edge_type, int64_t num_edges,
                      int64_t num_nodes, int64_t num_relations, void* stream,
                      rgcn_graph** out);
/* One direction between two node sets - the shard of a node-partitioned graph (SURVEY.md
 * section 8e): segments are key_node*R + rel over num_key_nodes (the rows this rank owns),
 * col[] ids refer to num_other_nodes (the gathered rows of all ranks).  edge_weight == NULL:
 * mean mode (divide by the segment size, i.e. the forward structure of a rank that holds all
 * in-edges of its rows); otherwise weighted-sum mode with the given per-edge weights in input
 * order (the transposed structure, weights = 1 / global cnt[dst, rel]).  Only direction 0 of
 * the returned handle exists (use transposed = 0). */
int rgcn_graph_create_bipartite(const int64_t* key_node, const int64_t* other_node,
                                const int64_t* edge_type, int64_t num_edges, int64_t num_key_nodes,
                                int64_t num_other_nodes, int64_t num_relations,
                                const float* edge_weight, void* stream, rgcn_graph** out);
void rgcn_graph_destroy(rgcn_graph* g);

/* sizes */
int64_t rgcn_graph_num_edges(const rgcn_graph* g);
int64_t rgcn_graph_num_nodes(const rgcn_graph* g);
int64_t rgcn_graph_num_relations(const rgcn_graph* g);
/* max over the segments of that direction of the sum of their |edge weights| (1 for the mean structure):
 * |aggregate row| <= bound * max |gathered table|.  0 if the direction does not exist. */
float rgcn_graph_weight_bound(const rgcn_graph* g, int transposed);
/* number of aggregate launches (tree levels) one rgcn_aggregate call issues */
int rgcn_graph_num_levels(const rgcn_graph* g, int transposed);

/* Relation occupancy of 32-row tiles: bit r of mask[t] is set iff some row of [32t, 32t+32)
 * has a non-empty (row, r) segment in that direction's structure; NULL when R > 32.  Typed
 * relations (PrimeKG: drug-gene edges never reach a disease node) leave whole row ranges of
 * agg exactly zero for a relation; passing this mask as `tile_mask` to the transforms below
 * lets them skip those all-zero tiles (results are unchanged: only 0 * w products are dropped).
 * Use the forward mask with rgcn_transform_fwd / _bwd_params (their A operand is the forward
 * aggregate) and the transposed mask with rgcn_transform_bwd_input.  Owned by the handle. */
const uint32_t* rgcn_graph_tile_mask(const rgcn_graph* g, int transposed, int64_t* num_row_tiles);

/* Device views of the bucketed arrays (for parity tests and sidecar files):
 *   rowptr int32[N*R+1], col int32[E], perm int64[E] (original column of each bucketed
 *   edge), val float32: cnt[N*R] = max(1, segment size) when transposed == 0,
 *   w_t[E] when transposed == 1.  The pointers stay owned by the handle. */
int rgcn_graph_arrays(const rgcn_graph* g, int transposed, const int32_t** rowptr,
                      const int32_t** col, const int64_t** perm, const float** val);
/* Same arrays copied (device to device, async on `stream`) into caller buffers of the sizes
 * above; any destination may be NULL. */
int rgcn_graph_export(const rgcn_graph* g, int transposed, int32_t* rowptr, int32_t* col,
                      int64_t* perm, float* val, void* stream);

/* Rebuild a handle from the arrays rgcn_graph_export wrote for both directions of a square
 * graph (SURVEY section 8f "next" row 3: the bucketed structure persisted next to the
 * reference's dict `.pt` graph file, `src/preprocess.py:256-261`, so that start-up skips the
 * sort).  All pointers are device pointers and are copied.  Index arrays are validated on the
 * device (rowptr non-decreasing from 0 to num_edges, col in [0, num_nodes), perm in
 * [0, num_edges)); RGCN_ERR_RANGE otherwise.  cnt / w_t are taken as given. */
int rgcn_graph_import(int64_t num_edges, int64_t num_nodes, int64_t num_relations,
                      const int32_t* rowptr, const int32_t* col, const int64_t* perm, const float* cnt,
                      const int32_t* rowptr_t, const int32_t* col_t, const int64_t* perm_t,
                      const float* w_t, void* stream, rgcn_graph** out);

/* A pending fixed-order reduction of parameter-gradient slabs (filled by
 * rgcn_transform_bwd_params_begin, consumed by rgcn_slab_reduce or rgcn_aggregate_and_reduce). */
typedef struct rgcn_slab_job {
  const float* slab;
  const float* bias_part;
  int32_t splits, K1, Kc, N;
  float* grad_weight;
  float* grad_root;
  float* grad_bias;
} rgcn_slab_job;

/* ------------------------------------------------------------------------------------
 * Gather + per-(node, relation) aggregation (rows A3 + A4, and their autograd, row A7).
 *
 *   transposed == 0:  agg[i*R + r, :] = (sum over edges e = (j -> i, r) of x[j, :]) / cnt[i, r]
 *                     (`index_select` + `scatter_add_` + count + clamp(min=1) + divide)
 *   transposed == 1:  agg[j*R + r, :] = sum over edges e = (j -> i, r) of x[i, :] * w_t[e]
 *
 * x: float[N, d] (ld = d), agg: float[N*R, d].  d % 4 == 0.  Segments with no edge are
 * written as exact zeros.  Summation inside a segment runs in the bucketed (= original
 * column) order for segments of <= 64 edges; longer ones are summed as a fixed tree (runs of
 * 64 edges, four runs to a pack, packs reduced in order), so the result is run-to-run
 * deterministic.  `workspace` holds one partial row per pack of the segments longer than 256
 * edges: rgcn_aggregate_workspace_bytes(g, t, d).
 * ---------------------------------------------------------------------------------- */
size_t rgcn_aggregate_workspace_bytes(const rgcn_graph* g, int transposed, int64_t d);
int rgcn_aggregate(const rgcn_graph* g, int transposed, const float* x, int64_t d, float* agg,
                   void* workspace, size_t workspace_bytes, void* stream);
/* rgcn_aggregate whose level-0 launch also performs a pending slab reduction (see
 * rgcn_transform_bwd_params_begin) as extra workgroups; `job` may be NULL.  For row widths the
 * gather splits over a second grid dimension (d > 256) the reduction is launched by itself first. */
int rgcn_aggregate_and_reduce(const rgcn_graph* g, int transposed, const float* x, int64_t d,
                              float* agg, void* workspace, size_t workspace_bytes,
                              const rgcn_slab_job* job, void* stream);
/* The same with the gathered table stored as IEEE fp16 (x_f16: half[N, d], d % 8 == 0) and fp32
 * accumulation / output: half the bytes per gathered row (BASELINE.json configs[4], "fp16
 * features + fp32 accumulate").  agg and the workspace stay fp32. */
int rgcn_aggregate_f16(const rgcn_graph* g, int transposed, const void* x_f16, int64_t d, float* agg,
                       void* workspace, size_t workspace_bytes, void* stream);
/* "amax buffer": DEVICE float[RGCN_AMAX_FLOATS]; its VALUE, max |tensor|, is the maximum over its 256 "heads"
 * (entries 0, 8, 16, ...; the other entries are never touched).  Kernels that produce a tensor publish wave
 * maxima into 64 of the heads, 128 bytes apart, with an atomic max on the bit pattern (order-free, so the value
 * is deterministic; spread so a launch's atomics do not queue on one address): every head must be zero before
 * such a producer runs (rgcn_absmax clears buffers on the side).  The split-precision transforms below scale
 * their operands by it. */
#define RGCN_AMAX_FLOATS 2048
/* rgcn_aggregate_and_reduce that also leaves max |agg| in the amax buffer `amax` (zeroed by the caller). */
int rgcn_aggregate_amax(const rgcn_graph* g, int transposed, const float* x, int64_t d, float* agg,
                        void* workspace, size_t workspace_bytes, const rgcn_slab_job* job, float* amax,
                        void* stream);
/* The gather WITHOUT its hub-tail launch: segments longer than 256 edges are left as partial rows in `workspace`
 * (which the caller keeps alive), to be summed by the split-precision transform that consumes the aggregate
 * (hub_graph / hub_partial arguments of rgcn_transform_fwd_split / _bwd_input_split) - one launch less per
 * gather.  deferrable: 1 if the structure has exactly one reduce level (every segment <= 131,072 edges) and d is
 * 64, 128 or 256.  job: as rgcn_aggregate_and_reduce (NULL: none). */
int rgcn_aggregate_deferrable(const rgcn_graph* g, int transposed, int64_t d);
int rgcn_aggregate_deferred(const rgcn_graph* g, int transposed, const float* x, int64_t d, float* agg,
                            void* workspace, size_t workspace_bytes, const rgcn_slab_job* job, void* stream);
/* One launch of the above (level in [0, rgcn_graph_num_levels)); calling the levels in order
 * equals rgcn_aggregate.  Lets a profiler bracket the level-0 gather kernel by itself. */
int rgcn_aggregate_level(const rgcn_graph* g, int transposed, int level, const float* x, int64_t d,
                         float* agg, void* workspace, size_t workspace_bytes, float* amax /* or NULL */,
                         void* stream);

/* ------------------------------------------------------------------------------------
 * Per-relation transform + root + bias (row A6), fp32 MFMA (v_mfma_f32_32x32x2_f32):
 *
 *   out[N, d_out] = sum_r agg[:, r, :] @ weight[r] + x @ root + bias
 *
 * agg: float[N, R*d_in], x: float[N, d_in], weight: float[R, d_in, d_out] (PyG layout, read in
 * place as the MFMA B operand), root: float[d_in, d_out] or NULL, bias: float[d_out] or NULL.
 * relu != 0 fuses the `F.relu` that follows conv1 (rgcn.py:124) into the epilogue:
 * out = max(out, 0).
 * ---------------------------------------------------------------------------------- */
int rgcn_transform_fwd(const float* agg, const float* x, const float* weight, const float* root,
                       const float* bias, int relu, const uint32_t* tile_mask, int64_t num_nodes,
                       int64_t num_relations, int64_t d_in, int64_t d_out, float* out, void* stream);

/* The forward transform on the fp16 matrix cores (BASELINE.json configs[4], "fp16 features + fp32
 * accumulate"): the same contraction with [agg | x] and [W ; root] rounded to IEEE fp16 (nearest
 * even) and fp32 accumulation (v_mfma_f32_32x32x16_f16); bias / ReLU / out in fp32.  agg and x are
 * still passed as fp32 (the backward, which stays fp32, needs them); `workspace` receives the
 * packed fp16 weight operand (rgcn_transform_fwd_f16_workspace_bytes).  d_in must be a multiple
 * of 32 (RGCN_ERR_UNSUPPORTED otherwise: use rgcn_transform_fwd). */
size_t rgcn_transform_fwd_f16_workspace_bytes(int64_t num_relations, int64_t d_in, int64_t d_out);
int rgcn_transform_fwd_f16(const float* agg, const float* x, const float* weight, const float* root,
                           const float* bias, int relu, const uint32_t* tile_mask, int64_t num_nodes,
                           int64_t num_relations, int64_t d_in, int64_t d_out, float* out,
                           void* workspace, size_t workspace_bytes, void* stream);

/* Autograd of A6 with respect to the layer input (row A7):
 *   grad_x[N, d_in] = sum_r gagg[:, r, :] @ weight[r]^T + g @ root^T
 * gagg: float[N, R*d_out] = rgcn_aggregate(transposed = 1) of g; g: float[N, d_out].
 * relu_mask (float[N, d_in] or NULL): when the layer's input x was produced by a fused-ReLU
 * layer, pass x itself; the epilogue then writes grad_x * (x > 0), i.e. the gradient with
 * respect to that producer's pre-activation (autograd of rgcn.py:124). */
int rgcn_transform_bwd_input(const float* gagg, const float* g, const float* weight,
                             const float* root, const float* relu_mask, const uint32_t* tile_mask,
                             int64_t num_nodes, int64_t num_relations, int64_t d_in, int64_t d_out,
                             float* grad_x, void* stream);

/* Autograd of A6 with respect to the parameters (row A7):
 *   grad_weight[r] = agg[:, r, :]^T @ g,  grad_root = x^T @ g,  grad_bias = colsum(g)
 * Split over node ranges into fp32 slabs that a second kernel sums in a fixed order
 * (deterministic; no float atomics).  grad_root / grad_bias may be NULL. */
size_t rgcn_transform_bwd_params_workspace_bytes(int64_t num_nodes, int64_t num_relations,
                                                 int64_t d_in, int64_t d_out);
int rgcn_transform_bwd_params(const float* agg, const float* x, const float* g,
                              const uint32_t* tile_mask, int64_t num_nodes, int64_t num_relations,
                              int64_t d_in, int64_t d_out, float* grad_weight, float* grad_root,
                              float* grad_bias, void* workspace, size_t workspace_bytes, void* stream);

/* The same in two halves, so that the short fixed-order slab reduction need not sit between two
 * launch boundaries: `_begin` launches the slab GEMM and describes the pending reduction in `*job`
 * (plain pointers into `workspace` and the three outputs; nothing is owned); the reduction then
 * either runs by itself (rgcn_slab_reduce) or rides as extra workgroups of the transposed gather
 * that follows in a layer's backward and does not depend on it (rgcn_aggregate_and_reduce).
 * `workspace` and the outputs must stay alive until that launch. */
int rgcn_transform_bwd_params_begin(const float* agg, const float* x, const float* g,
                                    const uint32_t* tile_mask, int64_t num_nodes, int64_t num_relations,
                                    int64_t d_in, int64_t d_out, float* grad_weight, float* grad_root,
                                    float* grad_bias, void* workspace, size_t workspace_bytes, void* stream,
                                    rgcn_slab_job* job);
int rgcn_slab_reduce(const rgcn_slab_job* job, void* stream);

/* ------------------------------------------------------------------------------------
 * The three transforms above (rows A6 / A7) on the fp16 matrix cores in SPLIT PRECISION: fp32 in,
 * fp32 out, 1e-5-class results at 3/16 of the fp32 MFMA's cycles.  Every operand value v is carried
 * as two fp16 numbers, v * 2^e = hi + lo with one power-of-two scale per operand tensor (its largest
 * magnitude scaled into [2^14, 2^15)), and a product is lo*hi + hi*lo + hi*hi in three
 * v_mfma_f32_32x32x16_f16 passes with fp32 accumulation (csrc/rgcn_transform_split.hip): ~2^-22
 * relative error per product, absolute error 2^-39 of the tensor maximum for elements far below it.
 *
 * The aggregate operand (agg / gagg) is scaled by a BOUND on its magnitude, agg_amax_mul * value(agg_amax):
 * pass the amax buffer of the table the aggregate was gathered FROM and the structure's
 * rgcn_graph_weight_bound (1 for a mean: a mean of rows cannot exceed the table's maximum) - no pass over
 * the aggregate, no atomics in the gather; or the aggregate's own maximum (rgcn_aggregate_amax) and 1.
 * The second operand (x / g) is scaled by its own maximum; the accumulator is carried from the one scale
 * to the other where the k loop passes between the operands (powers of two: exact).
 * *_amax arguments: an amax buffer (see rgcn_aggregate_amax) holding max |operand| as left by the operand's producer
 * (rgcn_aggregate_amax, rgcn_absmax, or the out_amax / grad_x_amax of a previous transform); NULL makes
 * the call scan that operand itself (one extra pass over it).  out_amax / grad_x_amax (or NULL):
 * receives max |result| (zeroed by the caller).  Shapes outside the kernels' tiling (d_in resp. d_out
 * not a multiple of 32; 64 for the parameter gradients) return RGCN_ERR_UNSUPPORTED: use the fp32 calls.
 * half != 0: ONE pass on the hi parts only - both operands rounded to fp16 under their per-tensor scales (loss
 * scaling per tensor), fp32 accumulate: the arithmetic of BASELINE.json configs[4] ("fp16 features + fp32
 * accumulate") for the three gradient GEMMs, 2^-11 relative per operand.
 * `workspace`: rgcn_transform_split_workspace_bytes (fwd, bwd_input),
 * rgcn_transform_bwd_params_split_workspace_bytes (bwd_params).
 * ---------------------------------------------------------------------------------- */
/* amax buffer `out` <- max |x[i]| in one launch, no atomics, no prior clearing (every head is written); the
 * same launch clears the heads of `zero_count` (<= 256) further amax buffers laid out back to back from
 * `zero_buffers` - the buffers the kernels of the coming pass publish into. */
int rgcn_absmax(const float* x, int64_t n, float* out, float* zero_buffers, int zero_count, void* stream);
/* The same for up to 8 tensors in the one launch, each into its own amax buffer: the embedding table AND the
 * layers' weights at the start of a pass.  tensors / numels / outs: HOST arrays of `count` entries. */
int rgcn_absmax_multi(int count, const float* const* tensors, const int64_t* numels, float* const* outs,
                      float* zero_buffers, int zero_count, void* stream);
/* The weights of one layer split ONCE per step for both transforms that multiply by them ([W ; root] as fp16
 * hi / lo images in the forward and in the input-gradient orientation - each k-contiguous and in MFMA
 * B-fragment order - and in [W ; root]'s own order, one scale): pass the result as `packed`
 * to the two calls below; with packed == NULL each call splits the weights itself (into its workspace). */
size_t rgcn_weights_split_bytes(int64_t num_relations, int64_t d_in, int64_t d_out);
int rgcn_weights_split_pack(const float* weight, const float* root, int64_t num_relations, int64_t d_in,
                            int64_t d_out, void* packed, size_t packed_bytes, void* stream);
/* Up to 4 layers in ONE launch; w_amax[l] / r_amax[l]: the amax buffers of weights[l] and roots[l] when a
 * previous launch (rgcn_absmax_multi) left them, else NULL arrays / entries (the kernel scans the weights itself).
 * All array arguments are HOST arrays of `count` entries. */
int rgcn_weights_split_pack_multi(int count, const float* const* weights, const float* const* roots,
                                  const int64_t* num_relations, const int64_t* d_in, const int64_t* d_out,
                                  const float* const* w_amax, const float* const* r_amax, void* const* packed,
                                  const size_t* packed_bytes, float* zero_buffers, int zero_count, void* stream);
/* zero_buffers / zero_count (round 4): `zero_count` contiguous amax buffers whose heads this launch clears on the
 * side, as rgcn_absmax does - so that it can BE the first launch of a pass whose operand maxima all come from the
 * optimizer (rgcn_adam_clip_step's amax_out): no scan of the input table, none of the weights. */
/* The first launch of a forward pass, as ONE launch: max |x| of the pass's input table into x_amax (zero_buffers /
 * zero_count as in rgcn_absmax) AND the split weights of up to 4 layers (as rgcn_weights_split_pack_multi without
 * given maxima: the kernel scans the weights itself).  Array arguments: HOST arrays of `count` entries. */
int rgcn_absmax_pack(const float* x, int64_t numel, float* x_amax, float* zero_buffers, int zero_count, int count,
                     const float* const* weights, const float* const* roots, const int64_t* num_relations,
                     const int64_t* d_in, const int64_t* d_out, void* const* packed, const size_t* packed_bytes,
                     void* stream);
size_t rgcn_transform_split_workspace_bytes(int64_t num_relations, int64_t d_in, int64_t d_out);
/* hub_graph / hub_transposed / hub_partial (NULL / 0 / NULL: the aggregate operand is complete): the operand came
 * from rgcn_aggregate_deferred over that structure and direction, with hub_partial its workspace - the transform
 * finishes the hub rows of each 64-row tile itself before reading it (and writes them into the aggregate, which
 * is therefore complete once the call has run).  Needs agg_amax / gagg_amax (an unfinished operand cannot be
 * scanned). */
int rgcn_transform_fwd_split(const float* agg, const float* x, const float* weight, const float* root,
                             const void* packed, const float* bias, int relu, const uint32_t* tile_mask,
                             int64_t num_nodes, int64_t num_relations, int64_t d_in, int64_t d_out,
                             const float* agg_amax, float agg_amax_mul, const float* x_amax, int half,
                             float* out, float* out_amax, void* workspace, size_t workspace_bytes, void* stream,
                             const rgcn_graph* hub_graph, int hub_transposed, float* hub_partial);
int rgcn_transform_bwd_input_split(const float* gagg, const float* g, const float* weight, const float* root,
                                   const void* packed, const float* relu_mask, const uint32_t* tile_mask,
                                   int64_t num_nodes, int64_t num_relations, int64_t d_in, int64_t d_out,
                                   const float* gagg_amax, float gagg_amax_mul, const float* g_amax, int half,
                                   float* grad_x, float* grad_x_amax, void* workspace, size_t workspace_bytes,
                                   void* stream, const rgcn_graph* hub_graph, int hub_transposed,
                                   float* hub_partial, float out_scale);
/* out_scale (> 0; 1 for none): grad_x is additionally multiplied by it in the epilogue (one rounding, before the
 * ReLU mask and before grad_x_amax is taken).  The two-layer encoder passes 1 / (1 - p) of the dropout between its
 * layers (src/models/rgcn.py:125) here: the dropped activations relu_mask = relu(z) * m / (1 - p) are positive
 * exactly where a unit is active AND kept, so mask and factor together are autograd of dropout(relu(z)) - without
 * rescaling (and re-splitting) the layer's weights every step. */
/* conv2's input gradient with conv1's transform-first product CHAINED behind it inside the workgroup (round 4; the backward of
 * conv1 -> ReLU -> conv2, /root/reference/src/models/rgcn.py:123-128): grad_x = gz = ([gagg | g] * [W_r^T ; root^T]) . relu_mask,
 * exactly as rgcn_transform_bwd_input_split (the same bits, grad_x_amax published alike), and t_out[N, (R1 + has_root1) * d_in1] =
 * gz * [W1_r^T | root1^T] from conv1's split weights `packed1` (its natural-order image) - what rgcn_transform_first_split(gz, ...)
 * would compute in a launch of its own, a launch that is almost all latency at K = d_in (four k-tiles).  The workgroup owns whole
 * rows of gz (d_in == 128, one column block), keeps its 64 x 128 tile as fp16 hi / lo fragments under the TILE's maximum (a
 * power-of-two scale per row tile factors out of every row's sum; where no lo part is subnormal the bits equal the per-tensor
 * scale's) and walks the column blocks of t_out with a four-slot LDS-DMA ring of conv1's weights.  _supported: d_in == 128.
 * Measured (MI355X, C2): 47.6-49.5 us for the two launches, 43.2 us chained. */
int rgcn_transform_bwd_input_chain_supported(int64_t num_relations, int64_t d_in, int64_t d_out, int64_t num_relations1,
                                             int64_t d_in1);
int rgcn_transform_bwd_input_chain_split(const float* gagg, const float* g, const float* weight, const float* root,
                                         const void* packed, const float* relu_mask, const uint32_t* tile_mask,
                                         int64_t num_nodes, int64_t num_relations, int64_t d_in, int64_t d_out,
                                         const float* gagg_amax, float gagg_amax_mul, const float* g_amax, float* grad_x,
                                         float* grad_x_amax, void* workspace, size_t workspace_bytes, void* stream,
                                         const rgcn_graph* hub_graph, int hub_transposed, float* hub_partial,
                                         float out_scale, const void* packed1, int has_root1, int64_t num_relations1,
                                         int64_t d_in1, float* t_out);
/* Transform-first half of the input gradient (layers with d_out >= 2 d_in): T[N, (R + 1) * d_in] =
 * g * [W_0^T | ... | W_{R-1}^T | root^T] from the split weights' natural-order image (no concatenation, no second
 * split); grad_x is then rgcn_aggregate over the merged transposed structure of T viewed [N * (R + 1), d_in].
 * workspace: >= 2 KB (partial maxima when g_amax is NULL). */
int rgcn_transform_first_split(const float* g, const void* packed, int has_root, int64_t num_nodes,
                               int64_t num_relations, int64_t d_in, int64_t d_out, const float* g_amax, int half,
                               float* t_out, void* workspace, size_t workspace_bytes, void* stream);
size_t rgcn_transform_bwd_params_split_workspace_bytes(int64_t num_nodes, int64_t num_relations,
                                                       int64_t d_in, int64_t d_out);
/* slab GEMM in split precision; the pending fixed-order reduction is consumed exactly like the one of
 * rgcn_transform_bwd_params_begin (rgcn_slab_reduce / rgcn_aggregate_and_reduce / rgcn_aggregate_amax) */
int rgcn_transform_bwd_params_split_begin(const float* agg, const float* x, const float* g,
                                          const uint32_t* tile_mask, int64_t num_nodes, int64_t num_relations,
                                          int64_t d_in, int64_t d_out, const float* agg_amax,
                                          float agg_amax_mul, const float* x_amax, const float* g_amax, int half,
                                          float* grad_weight,
                                          float* grad_root, float* grad_bias, void* workspace,
                                          size_t workspace_bytes, void* stream, rgcn_slab_job* job);

/* ------------------------------------------------------------------------------------
 * One layer forward as ONE kernel (rows A3 + A4 + A6 for the no-grad encoder; rgcn.py:123,128 under
 * evaluate.py's torch.no_grad()): out = [mean-aggregate(x) | x] * [W ; root] + bias (+ ReLU) with the aggregate
 * formed in LDS as the transform's A operand - no [N, R * d_in] tensor in HBM.  Mean structures only
 * (what rgcn_graph_create / _bipartite without weights bucket), split precision only; bit-identical to
 * rgcn_aggregate -> rgcn_transform_fwd_split.
 *   supported     1 if the kernel covers the shape: d_in in {64, 128, 256}, d_out in {128, 256},
 *                 num_relations < d_in / 2 and <= 32 (else use the two-call path)
 *   rowptr, col   int32 [N * R + 1], [rowptr[N * R]]: the CSR the kernel walks, in (node, relation) segment order
 *                 like the forward structure's (rgcn_graph_export), except that an id < 0 names row -id - 1 of
 *                 hub_agg ([*, d_in]) instead of a row of x.  Every segment LONGER than the inline limit the caller
 *                 chose (<= d_in / 4 edges) should be ONE such entry, its mean formed beforehand: the kernel walks a
 *                 segment edge by edge with one lane group per row, and a long walk makes its workgroup a
 *                 straggler.  The usual producer of hub_agg is rgcn_aggregate over a structure of the long segments
 *                 only (rgcn_graph_create_bipartite with key = hub row, one relation): same runs / packs / hub
 *                 reduce, same bits.  A segment's mean divides by its number of entries.
 *   tile_mask     rgcn_graph_tile_mask of the forward structure (relations no row of a 32-row block has are
 *                 skipped), or NULL
 *   packed        rgcn_weights_split_pack of this layer (the kernel reads its fragment-order forward images)
 *   x_amax        amax buffer of x (rgcn_absmax): scale of the whole A operand (a mean cannot exceed it)
 *   out_amax      optional amax buffer that receives max |out| (the next layer's x_amax)
 *   agg           NULL, or [N, R * d_in]: the aggregate is ALSO written there, every row of it (the training
 *                 forward keeps it for the parameter gradients; it is not read back here)
 * ---------------------------------------------------------------------------------- */
int rgcn_layer_fwd_fused_supported(int64_t num_relations, int64_t d_in, int64_t d_out);
int rgcn_layer_fwd_fused(const int32_t* rowptr, const int32_t* col, const uint32_t* tile_mask, int64_t num_nodes,
                         int64_t num_relations, const float* hub_agg, const float* x, const void* packed,
                         int has_root, const float* bias, int relu, int64_t d_in, int64_t d_out,
                         const float* x_amax, float* out, float* out_amax, float* agg, void* stream);

/* The input gradient of one layer the same way (row A7's dense half + the autograd of A3 + A4):
 *   grad_x = [transposed-aggregate(g) | g] * [W_r^T ; root^T]  (* (relu_mask > 0))
 * with the 1/cnt-weighted sums over out-edges formed in LDS - no [N, R * d_out] tensor in HBM.  rowptr_t / col_t /
 * w_t: the CSR of the TRANSPOSED structure (rgcn_graph_export(transposed = 1): segments (source, relation), ids =
 * destinations, w_t = 1 / cnt[destination, relation]) with the same convention for long segments: ONE entry, id =
 * -(row + 1) of hub_agg, weight 1, the row pre-aggregated by rgcn_aggregate over a weighted structure of the long
 * segments.  g_amax: amax buffer of g; gagg_amax_mul: rgcn_graph_weight_bound(g, 1) (the bound that scales the
 * weighted sums).  Shapes: d_out in {64, 128, 256}, d_in in {64, 128, 256}, num_relations < d_out / 2 and <= 32.
 * Bit-identical to rgcn_aggregate(transposed) -> rgcn_transform_bwd_input_split. */
int rgcn_layer_bwd_input_fused_supported(int64_t num_relations, int64_t d_in, int64_t d_out);
int rgcn_layer_bwd_input_fused(const int32_t* rowptr_t, const int32_t* col_t, const float* w_t,
                               const uint32_t* tile_mask_t, int64_t num_nodes, int64_t num_relations,
                               const float* hub_agg, const float* g, const void* packed, int has_root,
                               const float* relu_mask, int64_t d_in, int64_t d_out, const float* g_amax,
                               float gagg_amax_mul, float* grad_x, float* grad_x_amax, void* stream,
                               float out_scale /* as rgcn_transform_bwd_input_split */);

/* ------------------------------------------------------------------------------------
 * DistMult head (rows C1 + C2; rgcn.py:325-326 row gathers + rgcn.py:207-211):
 *
 *   scores[b] = sum_d H[hi(b), d] * Rm[ri(b), d] * T[ti(b), d]
 *
 * Each operand is a float matrix with row length d plus an optional int64 index vector;
 * a NULL index means "row b".  So (emb, head_idx), (emb, tail_idx), (rel_table, rel_idx)
 * is the fused gather form, and (head_emb, NULL), (tail_emb, NULL), (rel_rows, NULL) is
 * LinkPredictor.forward on already-gathered rows (e.g. after relation dropout).
 * ---------------------------------------------------------------------------------- */
int distmult_fwd(const float* h, const int64_t* h_idx, int64_t h_rows, const float* t, const int64_t* t_idx,
                 int64_t t_rows, const float* r, const int64_t* r_idx, int64_t r_rows, int64_t batch, int64_t d,
                 float* scores, void* stream);
/* *_rows: number of rows of each operand matrix.  An index outside [0, rows) (torch indexing would raise a
 * device-side assert, src/models/rgcn.py:325-326) is never dereferenced: it is clamped to row 0 and a sticky
 * per-device flag is raised, which rgcn_index_error_fetch reads (synchronising `stream`) and clears. */
int rgcn_index_error_fetch(int* host_flag, void* stream);

/* Backward, DETERMINISTIC: grad_h[hi(b), :] = sum over the samples b' with hi(b') == hi(b) of gs[b'] * r * t,
 * etc.  Duplicates in head / tail / relation ids are legal and frequent; no float atomics are used - every row
 * is summed by one wave in sample order (embedding rows: the row's first occurrence adds all of them; the
 * relation table: a fixed two-level tree over segments of >= 64 samples), so two runs give the same bits.
 * Rows reached through an index vector are WRITTEN; the rows nobody touches must read zero afterwards:
 * zero_tables != 0 - the first launch clears the indexed head / tail tables itself (all h_rows x d, t_rows x d
 * floats; extra workgroups of that launch, no fill launch of the caller's), zero_tables == 0 - the caller has
 * zeroed them.  A NULL index writes row b directly.  grad_h == grad_t (head and tail gathered from one table) is
 * one key space.  Any grad pointer may be NULL.  workspace: distmult_bwd_workspace_bytes(batch, d, r_idx ? r_rows : 0). */
size_t distmult_bwd_workspace_bytes(int64_t batch, int64_t d, int64_t r_rows);
int distmult_bwd(const float* grad_scores, const float* h, const int64_t* h_idx, int64_t h_rows, const float* t,
                 const int64_t* t_idx, int64_t t_rows, const float* r, const int64_t* r_idx, int64_t r_rows,
                 int64_t batch, int64_t d, float* grad_h, float* grad_t, float* grad_r, void* workspace,
                 size_t workspace_bytes, int zero_tables, void* stream);

/* The head fused with the training loss (SURVEY section 8f "next" row 1; reference
 * `self.criterion = nn.BCEWithLogitsLoss()` src/train.py:139 applied to the scores at
 * train.py:300): scores[b] as above and loss[b] = binary_cross_entropy_with_logits(scores[b],
 * labels[b]) per sample (the caller takes the mean).  labels: float[batch] in {0, 1}. */
int distmult_bce_fwd(const float* h, const int64_t* h_idx, int64_t h_rows, const float* t, const int64_t* t_idx,
                     int64_t t_rows, const float* r, const int64_t* r_idx, int64_t r_rows, const float* labels,
                     int64_t batch, int64_t d, float* scores, float* loss, void* stream);
/* Its backward: grad_scores[b] = grad_mean_loss[0] * (sigmoid(scores[b]) - labels[b]) / batch (autograd of
 * mean(loss)) formed inside the first kernel of distmult_bwd's deterministic accumulation.  grad_mean_loss is a
 * DEVICE pointer to one float (no host sync).  Workspace as distmult_bwd. */
int distmult_bce_bwd(const float* grad_mean_loss, const float* scores, const float* labels, const float* h,
                     const int64_t* h_idx, int64_t h_rows, const float* t, const int64_t* t_idx, int64_t t_rows,
                     const float* r, const int64_t* r_idx, int64_t r_rows, int64_t batch, int64_t d, float* grad_h,
                     float* grad_t, float* grad_r, void* workspace, size_t workspace_bytes, int zero_tables,
                     void* stream);

/* The step's bookkeeping around that loss in ONE launch (src/train.py:300, 321-326: the mean of BCEWithLogitsLoss,
 * `predictions = sigmoid(scores) > 0.5`, `correct += (predictions == labels).sum()`, `total_loss += loss.item() * n` -
 * nine elementwise / reduce launches when written in torch): mean_loss[0] = mean(loss[0..batch)) in a fixed order (two
 * runs give the same bits); loss_sum[0] += (double)mean * batch and correct[0] += #{b : (scores[b] > 0) == (labels[b] >
 * 0.5)} - the epoch's running sums, device-resident, either may be NULL; cursor[0] += cursor_add - the position of the
 * next batch in the epoch's permutation (rgcn_sample_batch reads it), NULL to leave it alone. */
int distmult_bce_reduce(const float* loss, const float* scores, const float* labels, int64_t batch, float* mean_loss,
                        double* loss_sum, int64_t* correct, int64_t* cursor, int64_t cursor_add, void* stream);

/* out[num_rows, d] = sum over b of rows[b, :] into row idx[b], deterministic (the two-level tree above by
 * itself): autograd of `table[idx]` for a table of few rows - LinkPredictor's relation embeddings when their
 * gathered rows go through dropout (src/models/rgcn.py:207-208). */
size_t rgcn_segment_sum_workspace_bytes(int64_t batch, int64_t d, int64_t num_rows);
int rgcn_segment_sum(const float* rows, const int64_t* idx, int64_t batch, int64_t d, int64_t num_rows, float* out,
                     void* workspace, size_t workspace_bytes, void* stream);

/* Mini-batch assembly on the device (SURVEY section 8f "next" row 1): the batch slice of the
 * shuffled train columns (src/train.py:223-245), NegativeSampler.sample (train.py:59-97) and the
 * positives + negatives + labels concatenation (train.py:281-288) in one launch.
 *   sample i <  batch : column order[cursor[0] + i] of (edge_index [2, E], edge_type [E]), label 1
 *   sample batch + j  : positive j / num_neg with its head (fair coin) or else its tail replaced
 *                       by a uniform node of [0, num_nodes), label 0
 * order: int64[E] (NULL = identity); cursor: DEVICE int64[1] (NULL = 0), so the launch can be
 * replayed from a HIP graph; rng: DEVICE int64[2] = {seed, epoch}: Philox4x32-10 keyed by the
 * seed, counter = (cursor*num_neg + j, epoch).  Outputs hold batch*(1+num_neg) entries.
 * Positions / columns outside [0, E) are clamped, never dereferenced. */
int rgcn_sample_batch(const int64_t* edge_index, const int64_t* edge_type, int64_t num_edges,
                      const int64_t* order, const int64_t* cursor, int64_t batch, int64_t num_neg,
                      int64_t num_nodes, const int64_t* rng, int64_t* heads, int64_t* tails,
                      int64_t* rels, float* labels, void* stream);

/* Gradient clipping + Adam / AdamW update of the training step (SURVEY section 8f "next" row 1;
 * reference `clip_grad_norm_(model.parameters(), grad_clip)` + `optimizer.step()`,
 * src/train.py:311-317) in two launches over a list of up to 32 fp32 parameter tensors:
 *   total = ||all grads||_2;  every gradient is scaled by min(1, max_norm / (total + 1e-6)) when
 *   max_norm > 0 (the scaled value is used, not written back); then torch.optim.Adam's update - or
 *   AdamW's decoupled decay when `adamw` != 0 - with bias correction from the step count.
 * params / grads / exp_avg / exp_avg_sq / steps: HOST arrays of `num_tensors` DEVICE pointers (the
 * pointers travel in the launch arguments, so a captured HIP graph keeps working on them); steps[t]:
 * DEVICE float[1], torch's per-tensor step count, bumped by one per call on the device.
 * total_norm: DEVICE float[1] or NULL.  workspace: rgcn_adam_workspace_bytes(num_tensors, numels).
 * amax_out (round 4; HOST array of `num_tensors` DEVICE amax buffers, entries or the array itself may be NULL): the
 * update leaves max |params[t]| AFTER the step in amax_out[t] (the heads it publishes into are cleared by the first
 * launch; every other entry must be zero and stays so) - the optimizer is the only writer of the embedding table and
 * of the layers' weights, so the next step's split-precision transforms take their operand scales from here instead
 * of scanning 8 MB + the weights at the head of their latency chain (rgcn_weights_split_pack_multi with maxima). */
size_t rgcn_adam_workspace_bytes(int num_tensors, const int64_t* numels);
int rgcn_adam_clip_step(int num_tensors, float* const* params, const float* const* grads,
                        float* const* exp_avg, float* const* exp_avg_sq, float* const* steps,
                        const int64_t* numels, float lr, float beta1, float beta2, float eps,
                        float weight_decay, int adamw, float max_norm, float* total_norm,
                        float* const* amax_out, void* workspace, size_t workspace_bytes, void* stream);

/* Tail ranking for evaluation (LinkPredictor.score_all_tails rgcn.py:215-243 +
 * compute_ranking_metrics evaluate.py:260-276, without materialising the [B, N] score matrix
 * or sorting it): hr = head_emb * rel_emb rows [B, d]; scores[b, n] = <hr[b], emb[n]> on the
 * fp32 MFMA; beaten_by[b] += #{ n != tail[b] : scores[b, n] > true_score[b] } (int atomics into
 * a caller-zeroed buffer), so rank[b] = 1 + beaten_by[b].  d must be a multiple of 32. */
int distmult_rank_tails(const float* hr, const float* emb, const float* true_score,
                        const int64_t* tail, int64_t batch, int64_t num_entities, int64_t d,
                        int32_t* beaten_by, void* stream);

/* The [B, num_entities] score matrix itself (LinkPredictor.score_all_tails rgcn.py:215-243: (h * r) @ E^T; the callers
 * that want every candidate's score - predict_all_tails, the top-k consumers - rather than a rank):
 * hr[b, :] = head[b, :] * rel[rel_idx[b], :] (rel_idx == NULL: rel holds one row per b), then
 * scores[b, n] = <hr[b], emb[n]> on the fp32 MFMA - the ranking launch's GEMM with a store epilogue, so a score here
 * and the score distmult_rank_tails compares are the same bits.  hr: caller's [B, d] buffer (the products, an output).
 * d must be a multiple of 32; a relation id outside [0, num_relations) yields a NaN row. */
int distmult_score_all_tails(const float* head, const float* rel, const int64_t* rel_idx, int64_t num_relations,
                             const float* emb, int64_t batch, int64_t num_entities, int64_t d, float* hr, float* scores,
                             void* stream);

/* Basis-decomposed relation weights (PyG RGCNConv with num_bases = B, BASELINE configs[2]:
 * `weight = (comp @ weight.view(num_bases, -1)).view(num_relations, in_channels, out_channels)`).
 * rgcn_basis_compose: weight[r, j] = sum_b comp[r, b] * basis[b, j], j over inner = in * out (a multiple of 4), b
 * ascending.  rgcn_basis_compose_bwd: grad_basis[b, j] = sum_r comp[r, b] * grad_weight[r, j] and grad_comp[r, b] =
 * sum_j grad_weight[r, j] * basis[b, j] - the latter as per-workgroup partial sums added in a fixed order (no float
 * atomics: two runs give the same bits); either output may be NULL.  workspace: rgcn_basis_compose_bwd_workspace_bytes
 * (needed for grad_comp only).  R * B <= 4096. */
int rgcn_basis_compose(const float* comp, const float* basis, int64_t R, int64_t B, int64_t inner, float* weight,
                       void* stream);
size_t rgcn_basis_compose_bwd_workspace_bytes(int64_t R, int64_t B, int64_t inner);
int rgcn_basis_compose_bwd(const float* grad_weight, const float* comp, const float* basis, int64_t R, int64_t B,
                           int64_t inner, float* grad_comp, float* grad_basis, void* workspace, size_t workspace_bytes,
                           void* stream);

/* ------------------------------------------------------------------------------------
 * A recorded pass issued by ONE call.  The reference re-runs its encoder from Python for every 1,024-edge batch
 * (src/train.py:291-297, src/models/rgcn.py:123,128); on a static graph the launches of a pass are the same list
 * every time and only the addresses of the step's tensors change.  `calls` names entry points of THIS header and
 * where their arguments sit in `args`; an argument is a constant, the bits of a double (float parameters), an
 * address relative to one of the caller's `bases` (the pass's buffers: its arena, its inputs), one of the call's
 * own rgcn_slab_job slots (filled by a *_begin call, consumed by a later one), the run's stream, or a HOST array
 * whose entries (constants / base-relative addresses) follow at args[index .. index + value).  The calls are
 * issued in order on `stream`; the first non-zero return code ends the run and is returned.  Nothing is
 * allocated, nothing synchronises: capturable like the calls it forwards to.
 * ---------------------------------------------------------------------------------- */
enum { RGCN_SEQ_IMM = 0, RGCN_SEQ_FLOAT = 1, RGCN_SEQ_BASE = 2, RGCN_SEQ_JOB = 3, RGCN_SEQ_STREAM = 4, RGCN_SEQ_ARRAY = 5 };
enum {
  RGCN_FN_ABSMAX = 0, RGCN_FN_ABSMAX_MULTI, RGCN_FN_ABSMAX_PACK, RGCN_FN_WEIGHTS_SPLIT_PACK_MULTI, RGCN_FN_AGGREGATE,
  RGCN_FN_AGGREGATE_AND_REDUCE, RGCN_FN_AGGREGATE_AMAX, RGCN_FN_AGGREGATE_DEFERRED, RGCN_FN_TRANSFORM_FWD_SPLIT,
  RGCN_FN_TRANSFORM_BWD_INPUT_SPLIT, RGCN_FN_TRANSFORM_FIRST_SPLIT, RGCN_FN_TRANSFORM_BWD_PARAMS_SPLIT_BEGIN,
  RGCN_FN_SLAB_REDUCE, RGCN_FN_LAYER_FWD_FUSED, RGCN_FN_LAYER_BWD_INPUT_FUSED, RGCN_FN_TRANSFORM_BWD_INPUT_CHAIN_SPLIT, RGCN_FN_COUNT
};
typedef struct rgcn_seq_arg {
  int32_t kind;  /* RGCN_SEQ_* */
  int32_t index; /* BASE: which base; JOB: which slot; ARRAY: first entry in args */
  int64_t value; /* IMM: the value; FLOAT: bits of a double; BASE: byte offset; ARRAY: number of entries */
} rgcn_seq_arg;
typedef struct rgcn_seq_call {
  int32_t fn;        /* RGCN_FN_* */
  int32_t num_args;  /* exactly the parameter count of that entry point */
  int64_t first_arg; /* index of its first argument in args */
} rgcn_seq_call;
int rgcn_sequence_run(const rgcn_seq_call* calls, int num_calls, const rgcn_seq_arg* args, int64_t num_args,
                      void* const* bases, int num_bases, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RGCN_HIP_H */
