/*
 * herald_amd.h -- C-ABI of the MI355X-native embedding-access engine.
 *
 * This is the drop-in boundary for Herald/Hetu's embedding hot path.  Two groups
 * of entry points are exported by libherald_amd.so:
 *
 *  (1) Reference-named operator symbols, with the reference's DLArray/DLStream
 *      calling convention, so the ctypes wrappers in python/hetu/gpu_links (ctypes) binds them
 *      unchanged.  Each declaration cites the reference prototype it replaces.
 *
 *  (2) `ha_*` plain-pointer entry points (device pointers, sizes, a hipStream_t
 *      passed as void*) that the reference-named symbols, the hetu_cache /
 *      laia_cache plugin mirrors and the sharded store are built from.
 *
 * Conventions (same as the reference, src/common/c_runtime_api.h + runtime_base.h):
 *   - return 0 on success, -1 on error (ha_last_error() holds the message);
 *   - caller owns every buffer; the callee borrows it for the call;
 *   - GPU work is asynchronous on the given stream; the caller synchronises;
 *   - indices at the operator boundary are float32 (python/hetu/dataloader.py:14),
 *     converted with (size_t)ids[i] semantics (src/dnnl_ops/EmbeddingLookup.cpp:31);
 *   - all embedding values are float32, row-major [rows, width].
 *
 * No torch types appear here; PyTorch is only used by the Python host side to
 * allocate device memory and provide streams.
 */
#ifndef HERALD_AMD_H_
#define HERALD_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- reference ABI structs (src/common/dlarray.h:18-60) ------------------ */
typedef enum { kCPU = 1, kGPU = 2 } DLDeviceType;
typedef struct {
    int device_id;
    DLDeviceType device_type;
} DLContext;
typedef struct {
    void *data;
    DLContext ctx;
    int ndim;
    int64_t *shape;
    int64_t *stride;
} DLArray;
typedef struct {
    int device_id;
    void *handle; /* points at a hipStream_t (reference: cudaStream_t*), may be NULL */
} DLStream;
typedef DLArray *DLArrayHandle;
typedef DLStream *DLStreamHandle;

typedef void *ha_stream_t; /* hipStream_t */

/* ---- library / error handling ------------------------------------------- */
const char *ha_version(void);
const char *ha_last_error(void);
/* Free the internal per-stream scratch used by the one-call reference-named ops. */
int ha_scratch_release(void);
/* Number of visible HIP devices (does not initialise a context beyond hipGetDeviceCount). */
int ha_device_count(void);

/* ========================================================================= *
 * (1) Reference-named operator symbols
 * ========================================================================= */

/* src/common/c_runtime_api.h:308-310, src/ops/EmbeddingLookup.cu:16-52
 * output[..., :] = input[(size_t)ids[...], :]                                */
int DLGpuEmbeddingLookUp(const DLArrayHandle input, const DLArrayHandle ids,
                         DLArrayHandle output, DLStreamHandle stream_handle);

/* src/common/c_runtime_api.h:312-314, src/ops/EmbeddingLookup.cu:75-132
 * input_grad = 0; input_grad[ids[i], :] += output_grad[i, :]  (dense result) */
int DLGpuEmbeddingLookUp_Gradient(const DLArrayHandle output_grad,
                                  const DLArrayHandle ids,
                                  DLArrayHandle input_grad,
                                  DLStreamHandle stream_handle);

/* src/common/c_runtime_api.h:569-571, src/ops/IndexedSlices.cu:17-48
 * output[indices[i], :] += values[i, :]                                      */
int IndexedSlicesOneSideAdd(const DLArrayHandle indices,
                            const DLArrayHandle values, DLArrayHandle output,
                            DLStreamHandle stream_handle);

/* src/common/c_runtime_api.h:700-702, src/ops/OptimizersSparse.cu:282-329
 * compressed[(int)inverse[i], :] += origin[i, :]   (compressed pre-zeroed by caller) */
int DeduplicateIndexedSlices(const DLArrayHandle origin,
                             const DLArrayHandle inverse,
                             DLArrayHandle compressed,
                             DLStreamHandle stream_handle);

/* src/common/c_runtime_api.h:704-706, src/ops/OptimizersSparse.cu:233-280
 * new_values[(int)indices[i], :] = values[i, :]   (new_values pre-zeroed by caller) */
int IndexedSlices2Dense(const DLArrayHandle values, const DLArrayHandle indices,
                        DLArrayHandle new_values, DLStreamHandle stream_handle);

/* src/common/c_runtime_api.h:645-648, src/ops/OptimizersSparse.cu:53-99
 * param[indices[i], :] -= lr * values[i, :]; duplicates applied in occurrence
 * order per row (bit-identical to cpu_SGDOptimizerSparseUpdate,
 * src/dnnl_ops/Optimizers.cpp:51-74).                                         */
int SGDOptimizerSparseUpdate(DLArrayHandle param,
                             const DLArrayHandle grad_indices,
                             const DLArrayHandle grad_values, float lr,
                             DLStreamHandle stream_handle);

/* Sparse optimizers on DEDUPLICATED indices (callers run IndexedSlices.deduplicate first,
 * python/hetu/gpu_links/OptimizerLink.py:60,78,95):
 * src/common/c_runtime_api.h:661-664 / src/ops/OptimizersSparse.cu:331-389 */
int AdaGradOptimizerSparseUpdate(DLArrayHandle param,
                                 const DLArrayHandle grad_indices,
                                 const DLArrayHandle grad_values,
                                 DLArrayHandle acc, float lr, float eps,
                                 DLStreamHandle stream_handle);
/* src/common/c_runtime_api.h:670-674 / src/ops/OptimizersSparse.cu:391-455 */
int AdamOptimizerSparseUpdate(DLArrayHandle param,
                              const DLArrayHandle grad_indices,
                              const DLArrayHandle grad_values,
                              DLArrayHandle expavg, DLArrayHandle expavgsq,
                              float lr, float beta1, float beta2, float beta1t,
                              float beta2t, float eps,
                              DLStreamHandle stream_handle);
/* src/common/c_runtime_api.h:681-686 / src/ops/OptimizersSparse.cu:457-522 */
int AdamWOptimizerSparseUpdate(DLArrayHandle param,
                               const DLArrayHandle grad_indices,
                               const DLArrayHandle grad_values,
                               DLArrayHandle expavg, DLArrayHandle expavgsq,
                               float lr, float beta1, float beta2,
                               float beta1t, float beta2t, float eps,
                               float weight_decay,
                               DLStreamHandle stream_handle);
/* src/common/c_runtime_api.h:639-643 / src/ops/OptimizersSparse.cu:3-51:
 * grad_values[i,:] += l2reg * param[grad_indices[i],:]  (deduplicated slices, OptimizerLink.py:15-20) */
int AddL2RegularizationSparse(const DLArrayHandle param,
                              const DLArrayHandle grad_indices,
                              DLArrayHandle grad_values, float l2reg,
                              DLStreamHandle stream_handle);
/* src/common/c_runtime_api.h:653-656 / src/ops/OptimizersSparse.cu:101-231: indices may repeat
 * (OptimizerLink.py:37-49 does not deduplicate).  velocity[id,:] += -lr*g per occurrence (Nesterov: the
 * parameter row too), in occurrence order instead of the reference's atomics; then the reference's DENSE
 * second phase over the whole array: param += velocity, velocity *= momentum (Nesterov: velocity *=
 * momentum first, then param += velocity). */
int MomentumOptimizerSparseUpdate(DLArrayHandle param,
                                  const DLArrayHandle grad_indices,
                                  const DLArrayHandle grad_values,
                                  DLArrayHandle velocity, float lr,
                                  float momentum, bool nesterov,
                                  DLStreamHandle stream_handle);
/* src/common/c_runtime_api.h:693-699 / src/ops/OptimizersSparse.cu:524-722 (deduplicated slices) */
int LambOptimizerSparseUpdate(DLArrayHandle param,
                              const DLArrayHandle grad_indices,
                              const DLArrayHandle grad_values,
                              DLArrayHandle expavg, DLArrayHandle expavgsq,
                              float lr, float beta1, float beta2, float beta1t,
                              float beta2t, float eps, float weight_decay,
                              DLStreamHandle stream_handle);
/* src/common/c_runtime_api.h:811-818, the reference's CPU operator names that python/hetu/_base.py:8-11,72
 * feature-probes.  Served by the HIP kernels (null stream, complete on return); there is no CPU arithmetic
 * behind them.  Arrays whose DLContext is the GPU are used where they lie.  HOST arrays (what the reference's
 * callers pass: EmbeddingLookUp.py:16-17, optimizer.py:203-207) are made visible to the device for the call:
 * below 64 MiB copied (H2D, kernel, D2H for the written one), from 64 MiB page-locked and mapped with
 * hipHostRegister -- the kernels then move only the rows the ids name across PCIe; the registration is kept
 * until ha_host_unmap(array) / ha_scratch_release() (callers pass the same parameter array every step).  An array
 * of that size must be unmapped BEFORE it is freed: an allocation that lands on the same address later would be
 * seen through the old mapping. */
int ha_host_unmap(void *host_array);
int cpu_EmbeddingLookup(const DLArrayHandle in_mat, const DLArrayHandle ids,
                        DLArrayHandle out_mat);
int cpu_SGDOptimizerSparseUpdate(DLArrayHandle param,
                                 const DLArrayHandle grad_indices,
                                 const DLArrayHandle grad_values, float lr);

/* ========================================================================= *
 * (2) Plain-pointer engine entry points.  All pointers are DEVICE pointers
 *     unless a parameter name ends in _host.
 * ========================================================================= */

/* Tolerance mode (process-wide, default off).  Off: every sparse update is the reference's serial chain, occurrence by
 * occurrence (cpu_SGDOptimizerSparseUpdate, src/dnnl_ops/Optimizers.cpp:65-72), bit for bit.  On: runs of 64 or more
 * occurrences of one key in a batch are applied as `row - tree_sum(lr * g)` in a fixed, deterministic order -- within
 * the 1e-5 relative BASELINE.json's north star allows for accumulated fp32 gradients; shorter runs stay bit-exact.
 * on = 1: one sixteen-wave tree over the whole run (oracle/qstep_model.py tree_coop).  on = 2: as 1, and the applies of a
 * FINISHED plan of 36,865 .. 2^20 ids on rows of up to 256 floats cut a run beyond 256 occurrences into chunks of 256 that
 * workgroups of their own sum, the chunk sums added in chunk order (tree_coop_chunked; the rule: listed_chunking) -- for
 * batches in which one key's run is the launch's critical path; at BASELINE configs[2]'s shape it is not (34.5 us without,
 * 35.8 with), so 1 is what bench.py's N>1 leg uses.  ha_get_tolerance_mode returns 0 / 1 / 2.
 * Read at launch time by ha_sgd_apply*, ha_push_apply*, ha_dedup_reduce*, ha_apply_mapped, ha_shard_serve_push and the
 * entry points built on them (not by ha_sgd_push_pull_* / ha_step_*, which are bit-exact throughout, nor by
 * ha_qstep_* / ha_qapply, which have their own documented tolerance classes). */
int ha_set_tolerance_mode(int on);
int ha_get_tolerance_mode(void);

/* ---- forward gather (replaces cpu_EmbeddingLookup / embedding_lookup_kernel) */
/* out[i,:] = table[(size_t)ids[i],:], ids float32.  width % 4 == 0 takes the
 * 16-byte vector path; any width >= 1 is supported.                          */
int ha_gather_f32ids(const float *table, int64_t rows, int64_t width,
                     const float *ids, int64_t n, float *out,
                     ha_stream_t stream);
/* Same with 64-bit keys (hetu_cache numpy-uint64 entry points, cache.cc:37-47). */
int ha_gather_u64ids(const float *table, int64_t rows, int64_t width,
                     const uint64_t *ids, int64_t n, float *out,
                     ha_stream_t stream);
int ha_gather_u32keys(const float *table, int64_t rows, int64_t width,
                      const uint32_t *keys, int64_t n, float *out,
                      ha_stream_t stream);

/* dst[(size_t)ids[i],:] = values[i,:] (ids deduplicated by the caller; the body of
 * IndexedSlices2Dense, src/ops/OptimizersSparse.cu:233-280). */
int ha_scatter_rows_f32ids(const float *values, const float *ids, int64_t n,
                           int64_t width, float *dst, int64_t rows,
                           ha_stream_t stream);

/* values[i] *= scale (one float32 rounding): the `values *= -lr` step of
 * ParameterServerCommunicateOp (python/hetu/gpu_ops/ParameterServerCommunicate.py:24,58-59). */
int ha_scale_f32(float *values, int64_t n, float scale, ha_stream_t stream);

/* ---- index plan: sorted-unique + inverse + counts + occurrence lists ------
 * Semantics of np.unique(ids, return_inverse=True, return_counts=True)
 * (python/hetu/ndarray.py:534,559) == hetu::Unique<T> (unqiue_tools.h:27-48)
 * == the std::map dedup of PSAgent::vecPushSparse (PSAgent.h:124-183):
 *   uniq   : ascending distinct keys
 *   inverse: inverse[i] = position of ids[i] in uniq
 *   counts : occurrences of each unique key
 *   perm   : stable argsort of ids (occurrence order inside a run of equal keys)
 *   seg    : seg[u] = first sorted position of unique key u; seg[U] = n
 * Keys are held as uint32 (tables up to 2^32-2 rows).
 *
 * A plan lives in a caller-provided device workspace of ha_plan_bytes(n) bytes
 * laid out as struct ha_plan_view describes; it is valid until the workspace is
 * reused.  n_unique is produced on the device (plan header) so that no host
 * sync is needed between plan and the kernels that consume it.
 * The first 256 bytes of a workspace (the header: n_unique and sticky flag
 * words) must be ZERO before its first use, and the zero fill must have
 * COMPLETED before the first call that names the workspace on another stream:
 * hipMemset / hipMemsetAsync on the null stream is not waited for by the host
 * and is not ordered with non-blocking streams -- synchronise the null stream
 * (or zero on the stream of the first call).  A late fill wipes n_unique.    */
typedef struct {
    int64_t n;           /* number of ids */
    int64_t *n_unique;   /* device scalar (first word of the workspace) */
    uint32_t *keys;      /* [n] ids converted to integer keys, original order */
    uint32_t *sorted;    /* [n] keys ascending */
    int32_t *perm;       /* [n] stable argsort: sorted[p] == keys[perm[p]] */
    int32_t *inverse;    /* [n] */
    uint32_t *uniq;      /* [n] first *n_unique valid */
    int32_t *counts;     /* [n] first *n_unique valid */
    int32_t *seg;        /* [n+1] first *n_unique+1 valid */
    int32_t *upos;       /* [n] upos[p] = unique index of sorted position p */
} ha_plan_view;

size_t ha_plan_bytes(int64_t n);
/* Fill `view` with the addresses of the sub-arrays of workspace `ws` (host-side
 * pointer arithmetic only; no device access). */
int ha_plan_view_of(void *ws, int64_t n, ha_plan_view *view);
/* Diagnostics: with HA_RADIX_STAMPS=1 in the environment the scatter launches of the radix sort (n > 36,864) stamp the GPU's
 * 100 MHz clock at their phase boundaries; out_host[24], eight slots per pass (tools/radix_phases.py). */
int ha_plan_radix_stamps(void *ws, int64_t n, uint64_t *out_host, ha_stream_t stream);
/* Build the plan from float32 ids / uint64 keys. */
int ha_plan_build_f32ids(const float *ids, int64_t n, void *ws,
                         ha_stream_t stream);
int ha_plan_build_u64ids(const uint64_t *ids, int64_t n, void *ws,
                         ha_stream_t stream);
/* Two-phase form.  ha_plan_sort_* produces only `keys`, `sorted` and `perm` -- all that
 * ha_sgd_apply / ha_push_apply need; ha_plan_finish adds n_unique, uniq, counts, seg, inverse and
 * upos (needed by ha_dedup_reduce, ha_plan_export_f32 and the cache).  build == sort + finish. */
int ha_plan_sort_u32keys(const uint32_t *keys, int64_t n, void *plan_ws, int key_bits, ha_stream_t stream);
int ha_plan_sort_f32ids(const float *ids, int64_t n, void *ws,
                        ha_stream_t stream);
int ha_plan_sort_u64ids(const uint64_t *ids, int64_t n, void *ws,
                        ha_stream_t stream);
/* The same with a known key range (key_limit = number of table rows, or any bound on the valid keys):
 * batches of 18,433 .. 36,864 ids then take the bucket sort -- one most-significant-digit scatter into
 * 2,048 ordered key ranges + stable rank-by-counting inside the ranges, three launches -- instead of the
 * O(n^2) single-launch sort or the three-pass LSD radix sort.  Keys at or above the limit are still sorted
 * correctly (they share the last range).  Identical results. */
/* The plans of `count` batches (ids[i], n[i] -> ws[i]; host arrays of device pointers): two launches for all of them
 * when every batch takes the rank-by-counting path (n <= 36,864 and no bucket sort), else one build per batch. */
int ha_plan_build_batch_f32ids_lim(const float *const *ids, const int64_t *n, void *const *ws, int count,
                                   uint64_t key_limit, ha_stream_t stream);
int ha_plan_build_batch_u64ids_lim(const uint64_t *const *ids, const int64_t *n, void *const *ws, int count,
                                   uint64_t key_limit, ha_stream_t stream);
/* the stable sorts alone of up to 16 small batches per launch (what ha_plan_sort_*_lim leaves in each workspace) */
int ha_plan_sort_batch_f32ids_lim(const float *const *ids, const int64_t *n, void *const *ws, int count,
                                  uint64_t key_limit, ha_stream_t stream);
int ha_plan_sort_batch_u64ids_lim(const uint64_t *const *ids, const int64_t *n, void *const *ws, int count,
                                  uint64_t key_limit, ha_stream_t stream);
int ha_plan_build_f32ids_lim(const float *ids, int64_t n, void *ws, uint64_t key_limit, ha_stream_t stream);
int ha_plan_sort_f32ids_lim(const float *ids, int64_t n, void *ws, uint64_t key_limit, ha_stream_t stream);
int ha_plan_build_u64ids_lim(const uint64_t *ids, int64_t n, void *ws, uint64_t key_limit, ha_stream_t stream);
int ha_plan_sort_u64ids_lim(const uint64_t *ids, int64_t n, void *ws, uint64_t key_limit, ha_stream_t stream);
int ha_plan_finish(void *ws, int64_t n, ha_stream_t stream);
/* keys already integer; key_bits = number of significant key bits (radix passes). */
int ha_plan_build_u32keys(const uint32_t *keys, int64_t n, void *ws,
                          int key_bits, ha_stream_t stream);
/* Export np.unique-style float32 results (what IndexedSlices.deduplicate hands to
 * DeduplicateIndexedSlices): uniq_f32[U], inverse_f32[n].  Either may be NULL. */
int ha_plan_export_f32(const void *ws, int64_t n, float *uniq_f32,
                       float *inverse_f32, ha_stream_t stream);

/* ---- backward: dedup-reduce and fused apply -------------------------------*/
/* reduced[u,:] = sum over occurrences i of key u, in occurrence order, starting
 * from 0.0f (IndexedSlices.cpu_deduplicate, python/hetu/ndarray.py:556-576;
 * PSAgent::vecPushSparse; Line::accumulate).  reduced must hold n rows; the
 * first *n_unique are written.  Deterministic, no atomics.                    */
int ha_dedup_reduce(const void *plan_ws, int64_t n, const float *grads,
                    int64_t width, float *reduced, ha_stream_t stream);

/* Same with every value multiplied by `scale` before it is summed: reduced[u,:] = sum(scale*g_i).
 * This is the worker side of a PS sparse push: `values *= -lr` (python/hetu/gpu_ops/
 * ParameterServerCommunicate.py:58-59) followed by the occurrence-order reduce of
 * PSAgent::vecPushSparse (PSAgent.h:146-160). */
int ha_dedup_reduce_scaled(const void *plan_ws, int64_t n, const float *grads,
                           int64_t width, float scale, float *reduced,
                           ha_stream_t stream);

/* table[key,:] -= lr * grads[i,:] for every occurrence i, applied per row in
 * occurrence order with separate multiply and subtract roundings: the result is
 * bit-identical to the serial loop of cpu_SGDOptimizerSparseUpdate
 * (src/dnnl_ops/Optimizers.cpp:65-72).  Each unique row is read and written once.
 * With lr = -1 this is the server side of sparse pushes applied one after the other in
 * occurrence order: table[key,:] = (table[key,:] + v_i0) + v_i1 ...  (PSFHandle.h:130-164). */
int ha_sgd_apply(float *table, int64_t rows, int64_t width,
                 const void *plan_ws, int64_t n, const float *grads, float lr,
                 ha_stream_t stream);
/* The same for a plan that is already FINISHED (ha_plan_build_*, or ha_plan_sort_* followed by ha_plan_finish): batches of
 * more than 36,864 ids map their waves to the unique keys the finish listed instead of to sorted positions (about a third
 * less time for 106,496 ids of 128-wide rows); smaller batches run as ha_sgd_apply.  Bit-identical results. */
int ha_sgd_apply_finished(float *table, int64_t rows, int64_t width, void *plan_ws, int64_t n, const float *grads,
                          float lr, ha_stream_t stream);

/* table[key,:] += scale * reduced_in_occurrence_order (PS push semantics:
 * worker pre-multiplies by -lr, server does +=;
 * ParameterServerCommunicate.py:58-59, PSFHandle.h:130-164).                  */
int ha_push_apply(float *table, int64_t rows, int64_t width,
                  const void *plan_ws, int64_t n, const float *grads,
                  ha_stream_t stream);

/* Mapped flavour used by the embedding cache: for every unique key u of a FINISHED plan
 *   dst[rowmap[u],:] = (dst_init[rowmap[u]] ? dst[rowmap[u],:] : 0) - lr*src[valmap[i0],:] - lr*src[valmap[i1],:] ...
 * over its occurrences i0 < i1 < ... (rowmap[u] < 0 skips u; NULL maps are identities).  With
 * lr = -1 this is Line::accumulate (src/hetu_cache/include/embedding.h:78-91) for a whole batch. */
int ha_apply_mapped(float *dst, int64_t dst_rows, int64_t width,
                    const void *plan_ws, int64_t n, const float *src, float lr,
                    const int32_t *rowmap, const int32_t *valmap,
                    const uint8_t *dst_init, ha_stream_t stream);
/* ha_apply_mapped into two destinations with one pass over src (the cache's Line::accumulate writes the
 * same ordered sum into a line's gradient buffer and into its data row, embedding.h:78-91):
 * dst as ha_apply_mapped with (rowmap, dst_init); dst2[rowmap2[u],:] always starts from its stored
 * value; rowmap2[u] < 0 = no second row.  Each result equals its own ha_apply_mapped call bit for bit. */
int ha_apply_mapped2(float *dst, int64_t dst_rows, float *dst2, int64_t width,
                     const void *plan_ws, int64_t n, const float *src, float lr,
                     const int32_t *rowmap, const int32_t *rowmap2,
                     const uint8_t *dst_init, ha_stream_t stream);

/* ---- fused deduplicate + optimizer step -----------------------------------------------------------
 * One call for what the reference does in two (python/hetu/gpu_links/OptimizerLink.py:52-100:
 * grad.deduplicate(stream), then {AdaGrad,Adam,AdamW}OptimizerSparseUpdate on the reduced slices): index plan
 * of the raw ids, then ONE launch in which every key's gradient rows are summed in occurrence order in
 * registers and consumed by that row's optimizer step -- the reduced slices never travel through HBM.
 * Bit-identical to ha_dedup_reduce + the optimizer symbol.  kind: 0 AdaGrad (state1 = accumulator, state2
 * unused), 1 Adam, 2 AdamW (state1 = exp_avg, state2 = exp_avg_sq).  hyper_host: 7 host floats
 * {lr, eps, beta1, beta2, beta1t, beta2t, weight_decay}.  plan_ws: ha_plan_bytes(n) of device scratch. */
int ha_sparse_opt_fused_f32ids(int kind, float *param, int64_t rows, int64_t width, const float *ids,
                               int64_t n, const float *grads, float *state1, float *state2,
                               const float *hyper_host, void *plan_ws, ha_stream_t stream);

/* ---- fused launches (two per training step) --------------------------------
 * ha_lookup_sort_*    == ha_gather_* + ha_plan_sort_*   in ONE launch (forward: the lookup of a
 *                        batch and the index plan its backward will need);
 * ha_sgd_apply_finish == ha_sgd_apply + ha_plan_finish  in ONE launch (backward);
 * ha_push_apply_finish likewise for the PS push semantics.  Bit-identical to the unfused calls;
 * batches larger than the single-launch sort limit fall back to the unfused sequence. */
int ha_lookup_sort_f32ids(const float *table, int64_t rows, int64_t width,
                          const float *ids, int64_t n, float *out,
                          void *plan_ws, ha_stream_t stream);
int ha_lookup_sort_u64ids(const float *table, int64_t rows, int64_t width,
                          const uint64_t *ids, int64_t n, float *out,
                          void *plan_ws, ha_stream_t stream);
int ha_sgd_apply_finish(float *table, int64_t rows, int64_t width,
                        void *plan_ws, int64_t n, const float *grads, float lr,
                        ha_stream_t stream);
int ha_push_apply_finish(float *table, int64_t rows, int64_t width,
                         void *plan_ws, int64_t n, const float *grads,
                         ha_stream_t stream);

/* ---- ONE launch per training step: backward of batch k beside the forward of batch k+1 -------------
 * ha_sgd_push_pull_* == ha_sgd_apply_finish(plan_cur, grads) followed by ha_lookup_sort_*(next_ids ->
 * next_out, plan_next), as ONE launch.  This is the reference's embedding_push_pull
 * (src/hetu_cache/src/cache.cc:356-422, python/hetu/cstable.py: push the gradients of one batch and
 * pull the rows of the next in one request) on the HBM-resident table; the rows returned for batch k+1
 * are the rows AFTER batch k's update, and every result is bit-identical to the two separate launches
 * (cpu_SGDOptimizerSparseUpdate, src/dnnl_ops/Optimizers.cpp:51-74, then cpu_EmbeddingLookup,
 * src/dnnl_ops/EmbeddingLookup.cpp:16-35).  The ids of the next batch are known one step ahead (the
 * reference prefetches too: ParameterServerCommunicate.py:96-139, dataloader.py:63-98).
 *
 * Rows that both batches touch are handed from the applying wave to the gathering wave inside the
 * launch through a per-batch PENDING TABLE (ha_pend_bytes() bytes of device memory, all-zero when
 * idle: ha_pend_reset).  The launch that SORTS a batch registers its keys in that batch's table
 * (ha_lookup_sort_pend_* for the first batch, ha_sgd_push_pull_* for every later one: pend_next); the
 * launch that APPLIES the batch drains it (pend_cur) and leaves it all-zero again.  So a plan sorted
 * with pending table P must be applied by ha_sgd_push_pull_* with pend_cur = P (n_next = 0 for the
 * last batch), and two tables alternate between consecutive batches.
 *   n_cur == 0: only the lookup + sort of the next batch (== ha_lookup_sort_pend_*);
 *   n_next == 0: only the apply + finish of the current batch.
 * grads / next_out must be 16-byte aligned.  The single launch needs rows that own their 128-byte lines
 * (width % 32 == 0, table 128-byte aligned) and batches of at most 18,432 ids (above that the bucket sort of the separate launches wins);
 * anything else runs as the separate launches with the same results (nothing is registered then).
 * Should a hand-off wait ever exceed its bound (~0.1 s; never observed), word 9 (int64) of
 * plan_next's 256-byte header is set to 1 (ha_plan_handoff_timeout) instead of hanging the device: the rows of
 * that lookup may then be stale, and the flag stays set.  The library never reads it back (that would put a
 * host synchronisation into every step); a caller that trains on the rows MUST poll it at a point where it
 * synchronises anyway (herald_amd.ops: IndexPlan.n_unique() raises, ops.check_handoff(plans) after a
 * synchronisation; examples/ctr/run_wdl.py checks it at every logging interval). */
size_t ha_pend_bytes(void);
int ha_pend_reset(void *pend, ha_stream_t stream);
int ha_lookup_sort_pend_f32ids(const float *table, int64_t rows, int64_t width,
                               const float *ids, int64_t n, float *out,
                               void *plan_ws, void *pend, ha_stream_t stream);
int ha_lookup_sort_pend_u64ids(const float *table, int64_t rows, int64_t width,
                               const uint64_t *ids, int64_t n, float *out,
                               void *plan_ws, void *pend, ha_stream_t stream);
int ha_sgd_push_pull_f32ids(float *table, int64_t rows, int64_t width,
                            void *plan_cur, int64_t n_cur, const float *grads, float lr,
                            void *pend_cur, const float *next_ids, int64_t n_next,
                            float *next_out, void *plan_next, void *pend_next,
                            ha_stream_t stream);
int ha_sgd_push_pull_u64ids(float *table, int64_t rows, int64_t width,
                            void *plan_cur, int64_t n_cur, const float *grads, float lr,
                            void *pend_cur, const uint64_t *next_ids, int64_t n_next,
                            float *next_out, void *plan_next, void *pend_next,
                            ha_stream_t stream);
/* development aid (tools/step_timeline.py): ha_sgd_push_pull_f32ids with per-wave time stamps,
 * dbg = device uint64[(number of workgroups) * 16 * 4] */
int ha_debug_step_timeline(float *table, int64_t rows, int64_t width,
                           void *plan_cur, int64_t n_cur, const float *grads, float lr,
                           void *pend_cur, const float *next_ids, int64_t n_next,
                           float *next_out, void *plan_next, void *pend_next,
                           unsigned long long *dbg, ha_stream_t stream);
/* device address of the sticky time-out flag (int64, 0 = fine) of a plan workspace: set by a launch whose wait for other
 * workgroups of the same launch exceeded its bound -- the hand-off above, or the histogram exchange of the one-launch radix
 * passes that sort more than 36,864 ids (all tiles of a pass are resident at once, so the wait is microseconds; the bound
 * keeps a scheduling surprise from hanging the device) */
int64_t *ha_plan_handoff_timeout(void *plan_ws);

/* The step with THREE batches of lookahead: ONE launch that
 *   - applies the gradients of batch k (plan_cur)
 *     == ha_sgd_apply(table, plan_cur, grads, lr)             [cpu_SGDOptimizerSparseUpdate, Optimizers.cpp:51-74]
 *   - writes the rows of batch k+1 AFTER that update to next_out (plan_next)
 *     == ha_gather_*(table, ids of batch k+1)                 [cpu_EmbeddingLookup, EmbeddingLookup.cpp:16-35]
 *   - finishes the plan of batch k+2 (plan_fin: unique keys, inverse, counts)   == ha_plan_finish
 *   - sorts batch k+3 (ahead_ids -> plan_ahead)               == ha_plan_sort_*(ahead_ids)
 * with results bit-identical to those calls.  Nothing waits inside the launch: batch k+1 was sorted and
 * finished by earlier calls, so the wave that holds the final values of a row of batch k writes them
 * straight to every output row of batch k+1 that names the key, and only the rows batch k does not touch
 * are copied from the table -- the rows both batches share (two thirds of the positions of consecutive
 * Criteo batches) are not read back from HBM.  The ids are known three batches ahead (the reference's
 * dataloader and laia scheduler run further ahead than that: dataloader.py:63-98, laia_scheduler.cc).
 *
 * Every batch owns a KEY TABLE (ha_step_tab_bytes() bytes of device memory, ha_step_tab_reset before first
 * use): filled by the call that finishes the batch's plan (tab_fin), read by the two calls after it
 * (tab_next, then tab_cur), and cleared by the call after those (tab_clear) -- four tables and four plans
 * rotate.  A stream of batches 0..B-1 is driven as calls c = -3 .. B-1 with cur = c, next = c+1, fin = c+2,
 * ahead = c+3 and n = 0 / null pointers for batches outside [0, B); call c uses tables T[c%4] (cur),
 * T[(c+1)%4] (next), T[(c+2)%4] (fin) and clears T[(c+3)%4].  Limits: at most ha_step_max_ids() ids per
 * batch, width % 4 == 0, table / grads / next_out 16-byte aligned; anything else is refused (-1): use
 * ha_sgd_push_pull_* or the separate calls. */
size_t ha_step_tab_bytes(void);
int64_t ha_step_max_ids(void);
int ha_step_tab_reset(void *tab, ha_stream_t stream);
int ha_step_f32ids(float *table, int64_t rows, int64_t width,
                   void *plan_cur, int64_t n_cur, const float *grads, float lr, const void *tab_cur,
                   void *plan_next, int64_t n_next, float *next_out, const void *tab_next,
                   void *plan_fin, int64_t n_fin, void *tab_fin,
                   const float *ahead_ids, int64_t n_ahead, void *plan_ahead,
                   void *tab_clear, ha_stream_t stream);
int ha_step_u64ids(float *table, int64_t rows, int64_t width,
                   void *plan_cur, int64_t n_cur, const float *grads, float lr, const void *tab_cur,
                   void *plan_next, int64_t n_next, float *next_out, const void *tab_next,
                   void *plan_fin, int64_t n_fin, void *tab_fin,
                   const uint64_t *ahead_ids, int64_t n_ahead, void *plan_ahead,
                   void *tab_clear, ha_stream_t stream);
/* development aid (tools/step_fwd_timeline.py): ha_step_f32ids with per-wave time stamps */
int ha_debug_step_fwd_timeline(float *table, int64_t rows, int64_t width,
                               void *plan_cur, int64_t n_cur, const float *grads, float lr, const void *tab_cur,
                               void *plan_next, int64_t n_next, float *next_out, const void *tab_next,
                               void *plan_fin, int64_t n_fin, void *tab_fin,
                               const float *ahead_ids, int64_t n_ahead, void *plan_ahead,
                               void *tab_clear, unsigned long long *dbg, ha_stream_t stream);

/* The same step driven by a WORK QUEUE that preparation launches have built (csrc/qstep.hip).
 *   ha_qapply        step c, ONE launch: applies the gradients of batch c and writes the rows of batch c+1 after that
 *                    update, item by item from QUEUE c: one wave per (unique key, column slice) applies the key's
 *                    occurrences to the row it holds in registers, writes the row back and writes it to every output
 *                    row of batch c+1 naming the key; keys only batch c+1 names are copies.  No probing, no waiting.
 *   ha_qplan_batch_* the plans of `count` batches, one workgroup each in one launch: unique keys, counts, segment
 *                    starts, inverse and occurrence lists (ascending inside every segment) of an index plan -- with
 *                    the unique keys in the order of a hash table's slots, NOT in key order (ha_plan_build_* gives
 *                    np.unique's order).
 *   ha_qqueue_batch  the queues of `count` steps, two workgroups each in one launch: queue k from the finished plans
 *                    of the batch step k applies (plans_a[k], n_a[k] ids; 0 = none) and of the batch it looks up
 *                    (plans_g[k]).
 * A plan takes one workgroup ~15 us and a queue ~13-20 us, longer than the items of a step (~11 us): callers prepare a
 * BLOCK of steps at a time on a stream of their own, beside the steps of the block before
 * (herald_amd.ops.QueueStepPipeline); ha_qprep_* (one plan and / or one queue) and ha_qstep_* (that, then the step: three
 * launches on one stream) are the serial forms.
 * Semantics and references as ha_step_* above (cpu_SGDOptimizerSparseUpdate, Optimizers.cpp:51-74;
 * cpu_EmbeddingLookup, EmbeddingLookup.cpp:16-35), with ONE difference: keys with fewer than 16 occurrences in a batch
 * take the reference's serial chain row = (row - lr*g0) - lr*g1 ... bit for bit; a key with 16 or more occurrences takes
 * row - T, T = a fixed (deterministic) tree sum of the lr*g_i -- within the 1e-5 relative that BASELINE.json's north star
 * allows for accumulated gradients, not bit-identical to the serial chain.  Integer results are exact.  Callers that need
 * the serial chain for every run length use ha_step_*.
 * A queue is ha_qstep_queue_bytes(n_cap, width) bytes (no initialisation); queue_n_cap = the n_cap it was sized with.
 * Limits: at most ha_qstep_max_ids() ids per batch, width % 4 == 0, table / grads / next_out 16-byte aligned.  ids / n /
 * plans / queues arguments of the batch calls are HOST arrays of `count` entries (device pointers inside).
 * CALLER REQUIREMENT (lookahead): step c reads queue c, which is built from the plans of batches c and c+1 -- the ids of
 * batch c+1 must be on the device before queue c is built, and the queue must be complete before ha_qapply(c) starts.
 * The serial forms need the ids 3 batches ahead; the block-pipelined schedule of ops.QueueStepPipeline (plans of block
 * b+2 and queues of block b+1 beside the steps of block b) needs them 3 * block batches ahead (48 at bench.py's block of
 * 16).  The reference's dataloader contract is one batch ahead in a 3-deep ring (python/hetu/dataloader.py:63-98) and
 * laia's queue is 5 deep: a caller that cannot name its ids that far ahead uses ha_step_* (3 batches) or
 * ha_sgd_push_pull_* (1 batch).
 * ha_qstep_init: per-device set-up (LDS attributes of the plan / queue kernels, the lane-order probe of the LDS atomics:
 * one small synchronous launch on the null stream), to be called once per device OUTSIDE any stream capture; the entry
 * points do the same lazily on their first call.  Returns 1 (atomic ranking), 0 (ballot ranking; HA_QSTEP_BALLOT=1) or -1. */
int ha_qstep_init(void);
int64_t ha_qstep_max_ids(void);
size_t ha_qstep_queue_bytes(int64_t n_cap, int64_t width);
int ha_qplan_batch_f32ids(const float *const *ids, const int64_t *n, void *const *plans, int64_t count,
                          ha_stream_t stream);
int ha_qplan_batch_u64ids(const uint64_t *const *ids, const int64_t *n, void *const *plans, int64_t count,
                          ha_stream_t stream);
int ha_qqueue_batch(int64_t rows, int64_t width, void *const *plans_a, const int64_t *n_a, void *const *plans_g,
                    const int64_t *n_g, void *const *queues, int64_t queue_n_cap, int64_t count, ha_stream_t stream);
int ha_qapply(float *table, int64_t rows, int64_t width, void *plan_cur, int64_t n_cur, const float *grads, float lr,
              void *plan_next, int64_t n_next, float *next_out, const void *queue_cur, int64_t queue_n_cap,
              ha_stream_t stream);
/* ha_qapply with the number of wave items of the queue (wave + copy items; -1 = unknown) when the caller knows it: the launch
 * then has no workgroups that find nothing.  A hint only: too small a number costs time (the waves loop over the items), never
 * correctness.  ha_qqueue_batch_counts = ha_qqueue_batch that also writes {wave items + 1, workgroup items + 1, copy items + 1}
 * of step k's queue to the first three of FOUR pinned host words counts_host[k] (0 = not built yet; the caller zeroes
 * them before the call): queues are built a block of steps ahead, so the host usually has the numbers when it enqueues
 * the step.  The fourth word is only ever written non-zero: bit 0 = the builder counted more items than the queue
 * holds (excluded by the layout's bounds), bit 1 = a plan's occurrence list was not in position order (the serial
 * chain's order; excluded on hardware whose LDS atomics the probe of ha_qstep_init accepted).  Either is a bug: the
 * caller must fail loudly (ops.QueueStepPipeline raises). */
int ha_qapply_sized(float *table, int64_t rows, int64_t width, void *plan_cur, int64_t n_cur, const float *grads,
                    float lr, void *plan_next, int64_t n_next, float *next_out, const void *queue_cur,
                    int64_t queue_n_cap, int64_t wave_items, ha_stream_t stream);
/* `count` consecutive steps enqueued by one call: per-step arrays of ha_qapply_sized's arguments (wave_items may be NULL). */
int ha_qapply_steps(float *table, int64_t rows, int64_t width, float lr, int64_t queue_n_cap, int64_t count,
                    void *const *plan_cur, const int64_t *n_cur, const float *const *grads, void *const *plan_next,
                    const int64_t *n_next, float *const *next_out, const void *const *queue_cur,
                    const int64_t *wave_items, ha_stream_t stream);
int ha_qqueue_batch_counts(int64_t rows, int64_t width, void *const *plans_a, const int64_t *n_a, void *const *plans_g,
                           const int64_t *n_g, void *const *queues, int64_t queue_n_cap, int64_t count,
                           uint32_t *const *counts_host, ha_stream_t stream);
/* Ordering the preparation stream and the apply's stream WITHOUT a packet on the apply's stream (an event record + an event
 * wait at every block boundary cost ~1 us per step at blocks of 16):
 *   (1) queue c must be complete before the apply of step c reads it: ha_qqueue_batch_epochs (= ha_qqueue_batch_counts +
 *       `epochs`, one non-zero tag per step) makes the builder finish every queue with a device-scope release and the tag;
 *       an apply launch that is handed the same tag (ha_qapply_steps_sync / ha_qapply_sync; epoch 0 = no check) looks at it
 *       before its first item -- it is there on the first look when the preparation runs a block ahead; otherwise the
 *       launch polls (~1 s at most), then raises *err = 8 (pinned host word) and does nothing.  The builder never waits
 *       for an apply launch.
 *   (2) the preparation must not rewrite plans / queues that steps still read: the last launch of a call completes
 *       `done_event` (an event of ha_event_create; it rides on that launch's own dispatch packet -- hipExtLaunchKernelGGL
 *       --, nothing is recorded between launches; with nothing to launch it is recorded on the stream), and the caller
 *       makes the preparation stream wait for it (ha_stream_wait_event).
 * The ids of the batches to plan must be complete on the device when the preparation is enqueued (nothing orders them
 * behind work of the apply's stream any more). */
void *ha_event_create(void);
int ha_event_destroy(void *event);
int ha_event_record(void *event, ha_stream_t stream);
int ha_stream_wait_event(ha_stream_t stream, void *event);
int ha_qqueue_batch_epochs(int64_t rows, int64_t width, void *const *plans_a, const int64_t *n_a, void *const *plans_g,
                           const int64_t *n_g, void *const *queues, int64_t queue_n_cap, int64_t count,
                           uint32_t *const *counts_host, const uint32_t *epochs, ha_stream_t stream);
int ha_qapply_steps_sync(float *table, int64_t rows, int64_t width, float lr, int64_t queue_n_cap, int64_t count,
                         void *const *plan_cur, const int64_t *n_cur, const float *const *grads, void *const *plan_next,
                         const int64_t *n_next, float *const *next_out, const void *const *queue_cur,
                         const int64_t *wave_items, const uint32_t *epochs, uint32_t *err, void *done_event,
                         ha_stream_t stream);
int ha_qapply_sync(float *table, int64_t rows, int64_t width, void *plan_cur, int64_t n_cur, const float *grads, float lr,
                   void *plan_next, int64_t n_next, float *next_out, const void *queue_cur, int64_t queue_n_cap,
                   int64_t wave_items, uint32_t epoch, uint32_t *err, void *done_event, ha_stream_t stream);
/* ha_qapply_steps_sync with, per step, the three pinned words the queue's builder wrote (ha_qqueue_batch_counts; NULL / zeros:
 * unknown): a launch that knows its queue's item counts is sized exactly, and its waves ask for their items without reading
 * the counts from the queue's header first (the epoch tag is then looked at beside the item). */
int ha_qapply_steps_counts(float *table, int64_t rows, int64_t width, float lr, int64_t queue_n_cap, int64_t count,
                           void *const *plan_cur, const int64_t *n_cur, const float *const *grads, void *const *plan_next,
                           const int64_t *n_next, float *const *next_out, const void *const *queue_cur,
                           const int64_t *wave_items, const uint32_t *const *counts_host, const uint32_t *epochs, uint32_t *err,
                           void *done_event, ha_stream_t stream);

/* The row / key exchange of the sharded sparse pull and push as RCCL point-to-point calls made by the LIBRARY on the caller's
 * stream (the reference's worker sends and receives inside C++ too: PSAgent::vecPullSparse / vecPushSparse,
 * ps-lite/include/ps/worker/PSAgent.h:124-237 -- U_s keys and U_s x d floats per server, ps/psf/sparse.h:9-32): one
 * ncclGroupStart / ncclSend + ncclRecv per peer with a non-zero count / ncclGroupEnd -- no Python in the step, no second
 * stream, capturable with the launches around it.  RCCL is resolved at run time (the copy the process has loaded, else
 * /opt/rocm's): ha_xchg_available() = 0 where there is none.  ha_xchg_unique_id: rank 0 makes the 128-byte id, the caller
 * carries it to the other ranks; ha_xchg_create: collective, one communicator per object, on the current device.  Counts are
 * per peer, buffers hold the peers' parts back to back in rank order; this rank's own counts are normally 0. */
int ha_xchg_available(void);
int ha_xchg_unique_id(void *id128);
void *ha_xchg_create(const void *id128, int world, int rank);
int ha_xchg_destroy(void *xchg);
int ha_xchg_bytes(void *xchg, const void *send, const int64_t *send_bytes, void *recv, const int64_t *recv_bytes,
                  ha_stream_t stream);
int ha_xchg_rows(void *xchg, const float *send, const int64_t *send_rows, float *recv, const int64_t *recv_rows, int64_t width,
                 ha_stream_t stream);

/* The WIDE path: batches of more than ha_qstep_max_ids() (7,168) and at most ha_qbig_max_ids() (131,072) ids -- BASELINE
 * configs[2] / configs[3]'s per-GPU shapes, 106,496 and 26,624 ids per step.  The batch is cut into hash buckets by one
 * stable multisplit (ha_qbig_buckets(n_cap) buckets of ~1,024 ids; a bucket may hold at most ha_qstep_max_ids() ids: hot
 * keys with more than ~6,000 occurrences together in one bucket overflow it -- the queues of the steps that touch such a
 * batch carry flag 4, header word and counts_host[k][3], and the caller runs those steps another way: ops.QueueStepPipeline
 * waits for counts_host[k][0] (written last, non-zero once queue k is complete) before it enqueues step k and falls back
 * to the sorted plan + ha_sgd_apply / ha_gather_*); every bucket is planned and joined like a narrow batch, side by side
 * (no sort anywhere), and
 * the apply is the narrow path's launch over all buckets' items.  ws: ha_qbig_plan_bytes(n_cap) bytes per batch (no
 * initialisation); queues: ha_qstep_queue_bytes(n_cap, width).  Semantics, tolerance classes and references as above
 * (Unique<T>'s contract -- every key once, occurrences in order -- unqiue_tools.h:27-48; cpu_deduplicate, ndarray.py:556-576).
 * ha_qbig_plan_batch_* = partition (two launches) + the bucket plans (one launch) of `count` batches; ha_qbig_queue_batch
 * = the queues of `count` steps (two launches); ha_qbig_apply = step c (one launch; coop_items = the queue's number of
 * workgroup items if known, counts_host[k][1] - 1, else -1; epoch / err / done_event as ha_qapply_sync, 0 / NULL / NULL
 * for none).  ha_qbig_plan_view: device pointers into a plan workspace for tests. */
int64_t ha_qbig_max_ids(void);
size_t ha_qbig_plan_bytes(int64_t n_cap);
int ha_qbig_buckets(int64_t n_cap);
int ha_qbig_plan_batch_f32ids(const float *const *ids, const int64_t *n, void *const *ws, int64_t n_cap, int64_t count,
                              ha_stream_t stream);
int ha_qbig_plan_batch_u64ids(const uint64_t *const *ids, const int64_t *n, void *const *ws, int64_t n_cap, int64_t count,
                              ha_stream_t stream);
int ha_qbig_queue_batch(int64_t rows, int64_t width, void *const *ws_a, const int64_t *n_a, void *const *ws_g,
                        const int64_t *n_g, void *const *queues, int64_t n_cap, int64_t count, uint32_t *const *counts_host,
                        const uint32_t *epochs, ha_stream_t stream);
int ha_qbig_apply(float *table, int64_t rows, int64_t width, void *ws_cur, int64_t n_cur, const float *grads, float lr,
                  void *ws_next, int64_t n_next, float *next_out, const void *queue_cur, int64_t n_cap, int64_t coop_items,
                  uint32_t epoch, uint32_t *err, void *done_event, ha_stream_t stream);
int ha_qbig_plan_view(void *ws, int64_t n_cap, void **boff, void **bhdr, void **uniq, void **counts, void **seg, void **gperm,
                      void **meta);

/* serial forms: ha_qprep_* = the plan of one batch (ahead_ids -> plan_ahead; n_ahead = 0: none) and / or the queue of one
 * step from (plan_a, plan_g) (n_a = n_g = 0: none); ha_qstep_*(call c) = ha_qprep_*(batch c+3; queue c+1 from plans
 * c+1 = plan_next, c+2 = plan_b1) followed by ha_qapply(step c).  A stream of batches 0..B-1 is driven as calls
 * c = -3 .. B-1 (n = 0 / null for batches outside [0, B)); four plans rotate and two queues alternate. */
int ha_qprep_f32ids(int64_t rows, int64_t width, const float *ahead_ids, int64_t n_ahead, void *plan_ahead,
                    void *plan_a, int64_t n_a, void *plan_g, int64_t n_g, void *queue_build, int64_t queue_n_cap,
                    ha_stream_t stream);
int ha_qprep_u64ids(int64_t rows, int64_t width, const uint64_t *ahead_ids, int64_t n_ahead, void *plan_ahead,
                    void *plan_a, int64_t n_a, void *plan_g, int64_t n_g, void *queue_build, int64_t queue_n_cap,
                    ha_stream_t stream);
int ha_qstep_f32ids(float *table, int64_t rows, int64_t width,
                    void *plan_cur, int64_t n_cur, const float *grads, float lr,
                    void *plan_next, int64_t n_next, float *next_out, const void *queue_cur,
                    void *plan_b1, int64_t n_b1, void *queue_build, int64_t queue_n_cap,
                    const float *ahead_ids, int64_t n_ahead, void *plan_ahead, ha_stream_t stream);
int ha_qstep_u64ids(float *table, int64_t rows, int64_t width,
                    void *plan_cur, int64_t n_cur, const float *grads, float lr,
                    void *plan_next, int64_t n_next, float *next_out, const void *queue_cur,
                    void *plan_b1, int64_t n_b1, void *queue_build, int64_t queue_n_cap,
                    const uint64_t *ahead_ids, int64_t n_ahead, void *plan_ahead, ha_stream_t stream);
/* a built queue's first words (device memory): {wave items, workgroup items, long, medium, small, copy items, copy
 * medium, copy small, overflow / order word of the apply part, overflow word of the copy part} */
const uint32_t *ha_qstep_queue_header(const void *queue);
/* development aids (tools/qstep_timeline.py): the items of a step with per-wave time stamps, dbg = device
 * uint64[(workgroups) * 4 * 4], zeroed; one plan + one queue with the phase stamps of their workgroups, ph = device
 * uint64[32], zeroed */
int ha_debug_qapply(float *table, int64_t rows, int64_t width, void *plan_cur, int64_t n_cur, const float *grads,
                    float lr, void *plan_next, int64_t n_next, float *next_out, const void *queue_cur,
                    int64_t queue_n_cap, unsigned long long *dbg, ha_stream_t stream);
int ha_debug_qprep_f32ids(int64_t rows, int64_t width, const float *ahead_ids, int64_t n_ahead, void *plan_ahead,
                          void *plan_a, int64_t n_a, void *plan_g, int64_t n_g, void *queue_build, int64_t queue_n_cap,
                          unsigned long long *ph, ha_stream_t stream);

/* A gate for a stream: the stream behind this call does not move on until the host has written a non-zero value to
 * *flag (pinned, device-visible host memory), or about 2 s have passed.  For callers that enqueue a whole sequence
 * behind the gate and release it once everything is queued (bench.py: the device must not wait for the host inside a
 * short timed region). */
int ha_stream_gate(const uint32_t *flag, ha_stream_t stream);
/* measurement aid (tools/occupy_ab.py): `wgs` workgroups of `threads` threads with `lds_bytes` of LDS that hold their wave
 * slots for `ticks` x 10 ns and touch no memory */
int ha_debug_occupy(int64_t wgs, int64_t threads, int64_t lds_bytes, int64_t ticks, ha_stream_t stream);

/* ha_sgd_apply_finish that also warms the memory-side cache for the NEXT batch: waves that have no
 * medium / long-run work (more than half of them) end by touching the table row that position p of
 * next_ids will gather, so the ha_lookup_sort_* / ha_gather_* that follows reads it from the
 * Infinity Cache instead of HBM.  The ids of the next batch are known one step ahead (the reference
 * prefetches too: ParameterServerCommunicate.py:96-139).  Results are those of ha_sgd_apply_finish. */
int ha_sgd_apply_finish_prefetch_f32ids(float *table, int64_t rows, int64_t width,
                                        void *plan_ws, int64_t n, const float *grads, float lr,
                                        const float *next_ids, int64_t next_n, ha_stream_t stream);

/* ---- row-range sharding (in-node replacement of the PS/worker split) -------
 * For the plan's sorted unique keys: offsets[g] = index of the first unique key owned by shard g
 * (offsets[nshard] = n_unique) and local_keys[u] = uniq[u] - starts[owner(u)], where shard g owns
 * rows [starts[g], starts[g+1]) -- AveragePartitioner ranges (ps-lite/include/ps/partitioner.h:46-57)
 * routed like PSAgent (PSAgent.h:537-560).  starts_host[nshard+1] is a HOST array; offsets
 * (int32[nshard+1]) and local_keys (uint32[n]) are device arrays. */
int ha_shard_bucket(const void *plan_ws, int64_t n, const int64_t *starts_host,
                    int nshard, int32_t *offsets, uint32_t *local_keys,
                    ha_stream_t stream);

/* One call for the routing of a batch: ha_plan_build_* of the ids, then the shard-local keys as
 * ha_shard_bucket writes them, and meta[0] = n_unique, meta[1+g] = number of unique keys owned by
 * shard g (device int64[1+nshard]: the send counts of the counts all-to-all and of the host read-back). */
int ha_shard_route_f32ids(const float *ids, int64_t n, void *plan_ws,
                          const int64_t *starts_host, int nshard, int64_t *meta,
                          uint32_t *local_keys, ha_stream_t stream);
int ha_shard_route_u64ids(const uint64_t *ids, int64_t n, void *plan_ws,
                          const int64_t *starts_host, int nshard, int64_t *meta,
                          uint32_t *local_keys, ha_stream_t stream);

/* The routing of a batch as ONE message per owner: ha_plan_build_* of the ids, then
 * send[g * (1 + cap) + 0] = number of this rank's unique keys that shard g owns, followed by those keys as
 * shard-local offsets (PSAgent.h:537-560), in a fixed frame of 1 + cap int32 per owner (cap >= n), and
 * meta[0] = n_unique, meta[1+g] = the same counts (device int64[1+nshard]) for the host read-back that sizes the
 * row exchanges.  The frames of all ranks travel in one equal-split all-to-all -- the reference sends key
 * lists and their lengths in one message too (PSAgent::vecPullSparse, PSAgent.h:185-237).
 * ha_shard_route_unpack turns the received frames recv[g] into the W key lists concatenated in rank order
 * (keys_out: what ha_gather_u32keys / ha_shard_serve_push take) and recv_cnt[g] (device int64[nshard]). */
int ha_shard_route_pack_f32ids(const float *ids, int64_t n, void *plan_ws,
                               const int64_t *starts_host, int nshard, int64_t cap,
                               int64_t *meta, int32_t *send, ha_stream_t stream);
int ha_shard_route_pack_u64ids(const uint64_t *ids, int64_t n, void *plan_ws,
                               const int64_t *starts_host, int nshard, int64_t cap,
                               int64_t *meta, int32_t *send, ha_stream_t stream);
int ha_shard_route_unpack(const int32_t *recv, int nshard, int64_t cap, int64_t *recv_cnt,
                          uint32_t *keys_out, ha_stream_t stream);

/* The same routing with FIXED frames, for a step whose launches and exchange sizes do not depend on device-side
 * counts (it replays from a hipGraph; PSAgent::vecPullSparse / vecPushSparse, PSAgent.h:124-237, with the next
 * batch routed ahead as ParameterServerCommunicate.py:147-185 prefetches it).  Per owner g:
 *   send[g * (2 + rcap)] = { count_g, overflow, rcap shard-local keys, 0xFFFFFFFF beyond count_g }
 * `overflow` = this batch names more than rcap unique keys of SOME owner (the same word in every frame of the sender:
 * after the equal-split key exchange every rank knows whether any rank overflowed and takes the sized exchange above
 * for that batch).  rowmap[u] = frame slot g * rcap + j of the plan's unique key u (-1 beyond rcap): the row map
 * ha_apply_mapped takes to reduce the gradients of a batch straight into the push frames; posmap[i] = frame slot
 * of position i (nshard * rcap, a zero row, beyond rcap): the key list ha_gather_u32keys takes to expand the
 * pulled row frames to the positions.  ha_shard_frames_unpack: received key frames -> keys_fixed[nshard * rcap]
 * (rank order, 0xFFFFFFFF in unused slots: a zero row for ha_gather_u32keys, skipped by ha_shard_serve_push) and
 * state = { any rank overflowed, keys received } (device int32[2]).  ha_shard_frames_pack is the second half of
 * ha_shard_frames_route_* for a plan that is already built. */
int ha_shard_frames_route_f32ids(const float *ids, int64_t n, void *plan_ws, const int64_t *starts_host,
                                 int nshard, int64_t rcap, int64_t frame_stride, int32_t *send, int32_t *rowmap,
                                 int32_t *posmap, ha_stream_t stream);
int ha_shard_frames_route_u64ids(const uint64_t *ids, int64_t n, void *plan_ws, const int64_t *starts_host,
                                 int nshard, int64_t rcap, int64_t frame_stride, int32_t *send, int32_t *rowmap,
                                 int32_t *posmap, ha_stream_t stream);
int ha_shard_frames_pack(const void *plan_ws, int64_t n, const int64_t *starts_host, int nshard, int64_t rcap,
                         int64_t frame_stride, int32_t *send, int32_t *rowmap, int32_t *posmap, ha_stream_t stream);
int ha_shard_frames_unpack(const int32_t *recv, int nshard, int64_t rcap, int64_t frame_stride, uint32_t *keys_fixed,
                           int32_t *state, ha_stream_t stream);
/* A block of batches in one launch each (the plans by ha_plan_build_batch_*): batch i's frames start
 * i * (2 + rcap) words into every owner's stride, rowmap / posmap / keys_fixed / state are per-batch arrays of `count`
 * device pointers given on the host. */
int ha_shard_frames_pack_batch(const void *const *plan_ws, const int64_t *n, int count, const int64_t *starts_host,
                               int nshard, int64_t rcap, int64_t frame_stride, int32_t *send, int32_t *const *rowmap,
                               int32_t *const *posmap, ha_stream_t stream);
int ha_shard_frames_unpack_batch(const int32_t *recv, int count, int nshard, int64_t rcap, int64_t frame_stride,
                                 uint32_t *const *keys_fixed, int32_t *const *state, ha_stream_t stream);
/* frame_stride: int32 words between the frames of consecutive owners, >= 2 + rcap.  With 2 + rcap the frames of one
 * batch are contiguous; a larger stride interleaves the frames of SEVERAL batches per owner, [owner][batch][2 + rcap],
 * so that the key frames of a block of batches travel in one equal-split all-to-all (herald_amd/sharded.py routes a
 * block of batches at a time, ahead of the steps that use them).
 * ha_shard_frames_serve_pull: ha_shard_frames_unpack and the owner-side gather of a pull in ONE launch
 * (PSHandler::serve(SparsePull), ps-lite/include/ps/server/PSFHandle.h:101-128): rows_out[g * rcap + j, :] =
 * table[key j of rank g, :] for the live slots (unused slots are not written), keys_fixed and state as
 * ha_shard_frames_unpack.  `state` may be pinned host memory: the host then reads the overflow word without a copy
 * in the stream. */
int ha_shard_frames_serve_pull(const float *table, int64_t rows, int64_t width, const int32_t *recv, int nshard,
                               int64_t rcap, int64_t frame_stride, float *rows_out, uint32_t *keys_fixed,
                               int32_t *state, ha_stream_t stream);

/* SIZED frames: the same key frames, routed a block of batches ahead, also leave the per-owner counts of every batch on
 * the device and in pinned host memory, so that the two row exchanges of a step carry exactly the rows the batch names
 * (the reference's messages hold U_s keys and U_s x d floats per server: PSAgent::vecPullSparse / vecPushSparse,
 * ps-lite/include/ps/worker/PSAgent.h:167-172,217-226; ps/psf/sparse.h:9-32) without a host read-back in the step, and
 * the keys a rank owns itself never enter an exchange.
 *   meta (int32[2 + 2 nshard]; meta_dev on the device, meta_host in pinned host memory, same contents):
 *     [0] some rank overflowed its key frames (the batch then takes the sized exchange with a read-back, as above)
 *     [1] keys received   [2 + g] unique keys this rank names of owner g   [2 + nshard + g] keys rank g names of this rank
 *   ha_shard_frames_pack_batch_sized   = ha_shard_frames_pack_batch + the send counts and the SIZED maps (self = this rank):
 *     posmap[i] = 0x80000000 | shard-local key  if position i names a key of this rank (the expand reads the table itself)
 *               = index of its unique key among the unique keys of the OTHER owners in key order (= the row of the
 *                 received rows: the all-to-all concatenates the owners' answers in rank order)
 *               = 0xFFFFFFFF (a zero row) for a batch that overflowed
 *     rowmap[u] = the same compact index for a key of another owner (region A of the push buffer), nshard * rcap + j for
 *                 the j-th key of this rank's own range (region S); -1 for a batch that overflowed
 *   ha_shard_frames_unpack_batch_sized = ha_shard_frames_unpack_batch + the receive counts; meta_host[0] is written last
 *   ha_shard_sized_serve_pull  rows_out[roff(g) + j,:] = table[key j of rank g,:] for the other ranks' requests, roff(g) =
 *     sum of the receive counts of the ranks before g without this rank (PSHandler::serve(SparsePull), PSFHandle.h:101-128)
 *   ha_gather2_u32map          the expand: out[i,:] = table[map & 0x7FFFFFFF,:] if map[i] has bit 31, else recv[map[i],:]
 *   ha_shard_sized_serve_push  push_buf = (2 nshard + 1) * rcap rows: A = rows for the other owners [0, nshard rcap), S =
 *     this rank's own keys [nshard rcap, +rcap), B = the rows received, compact in rank order [(nshard + 1) rcap, ...);
 *     the nshard key lists are merged in rank order and table[key,:] = ((table[key,:] + v_0) + v_1) ... (PSFHandle.h:130-164);
 *     total = keys received (host-known); plan_ws: ha_plan_bytes(nshard * rcap) of scratch
 *   ha_push_apply_scaled_finished  a rank alone (nshard = 1): reduce and server add of its own keys in ONE launch. */
int ha_shard_frames_pack_batch_sized(const void *const *plan_ws, const int64_t *n, int count, const int64_t *starts_host,
                                     int nshard, int self, int64_t rcap, int64_t frame_stride, int32_t *send,
                                     int32_t *const *rowmap, int32_t *const *posmap, int32_t *const *meta_dev,
                                     int32_t *const *meta_host, ha_stream_t stream);
int ha_shard_frames_unpack_batch_sized(const int32_t *recv, int count, int nshard, int64_t rcap, int64_t frame_stride,
                                       uint32_t *const *keys_fixed, int32_t *const *meta_dev, int32_t *const *meta_host,
                                       ha_stream_t stream);
int ha_shard_sized_serve_pull(const float *table, int64_t rows, int64_t width, const uint32_t *keys_fixed, int nshard,
                              int self, int64_t rcap, const int32_t *meta_dev, float *rows_out, ha_stream_t stream);
int ha_gather2_u32map(const float *table, int64_t rows, const float *recv, int64_t recv_rows, int64_t width,
                      const uint32_t *map, int64_t n, float *out, ha_stream_t stream);
int ha_shard_sized_serve_push(float *table, int64_t rows, int64_t width, const uint32_t *keys_fixed, int nshard, int self,
                              int64_t rcap, const int32_t *meta_dev, int64_t total, const float *push_buf, void *plan_ws,
                              ha_stream_t stream);
int ha_push_apply_scaled_finished(float *table, int64_t rows, int64_t width, void *plan_ws, int64_t n, const float *grads,
                                  float scale, ha_stream_t stream);

/* The sharded step as ONE native call: what PSAgent::vecPullSparse / vecPushSparse do inside one C++ call each
 * (ps-lite/include/ps/worker/PSAgent.h:124-237) -- the step's launches and its two row exchanges enqueued on the caller's
 * stream by the library, sized from the per-owner counts the routing left in pinned host memory a block ago.
 * ha_shard_slot describes one routed batch (herald_amd/sharded.py fills it once per routing slot):
 *   counts_host[2 + g] = unique keys this rank names at owner g, [2 + world + g] = unique keys rank g names here (pinned; the
 *   routing block's event has completed -- the caller looked at the overflow word [0] behind it);
 *   keys_fixed / meta_dev / posmap / rowmap: as ha_shard_frames_*_sized left them; plan_ws: the batch's index plan;
 *   owner_plan_ws: ha_plan_bytes(world * rcap + 16) of scratch for the owner-side merge (world > 1).
 * _pull: owner gather -> rows exchange (ha_xchg_rows on `xchg`) -> positions from the own shard / the received rows into out;
 * _push: reduce of scale * values by unique key -> rows exchange -> rank-ordered merge + apply; world 1: the lookup and one
 * reduce-and-add launch, xchg may be NULL.  ha_shard_step = both; ha_shard_steps = `count` steps (slots / outs / values per
 * step).  pull_send / pull_recv: [pull_rows = world * rcap, width]; push_buf: [push_rows = (2 world + 1) rcap, width] (rows
 * for the other owners | own keys | rows received); zero_flags: push_rows zero bytes. */
typedef struct ha_shard_slot {
    int32_t world, rank;
    int64_t rcap, n;
    void *plan_ws;
    const uint32_t *keys_fixed;
    const int32_t *meta_dev;
    const uint32_t *posmap;
    const int32_t *rowmap;
    const int32_t *counts_host;
    void *owner_plan_ws;
} ha_shard_slot;
int ha_shard_step_pull(const float *table, int64_t rows, int64_t width, const ha_shard_slot *slot, void *xchg, float *pull_send,
                       float *pull_recv, int64_t pull_rows, float *out, ha_stream_t stream);
int ha_shard_step_push(float *table, int64_t rows, int64_t width, const ha_shard_slot *slot, void *xchg, float *push_buf,
                       int64_t push_rows, const uint8_t *zero_flags, const float *values, float scale, ha_stream_t stream);
int ha_shard_step(float *table, int64_t rows, int64_t width, const ha_shard_slot *slot, void *xchg, float *pull_send,
                  float *pull_recv, int64_t pull_rows, float *push_buf, int64_t push_rows, const uint8_t *zero_flags, float *out,
                  const float *values, float scale, ha_stream_t stream);
int ha_shard_steps(float *table, int64_t rows, int64_t width, int64_t count, const ha_shard_slot *const *slots, void *xchg,
                   float *pull_send, float *pull_recv, int64_t pull_rows, float *push_buf, int64_t push_rows,
                   const uint8_t *zero_flags, float *const *outs, const float *const *values, float scale, ha_stream_t stream);

/* Owner side of a sparse push (PSHandler::serve(SparsePush), ps-lite/include/ps/server/PSFHandle.h:130-164):
 * table[keys[j],:] = (table[keys[j],:] + values[a,:]) + values[b,:] ... over the positions a < b < ... that
 * list the key, i.e. in list order (the W received sorted lists concatenated in rank order).
 * plan_ws: ha_plan_bytes(n) of scratch. */
int ha_shard_serve_push(float *table, int64_t rows, int64_t width, const uint32_t *keys,
                        int64_t n, const float *values, void *plan_ws, ha_stream_t stream);

/* ha_shard_serve_push for the received frames of a framed push: keys_fixed = nshard lists of rcap slots in rank order,
 * each ascending with 0xFFFFFFFF in the unused slots (ha_shard_frames_unpack), values[nshard * rcap, width] the row
 * frames.  The lists are merged (stable: equal keys in rank order) instead of sorted; plan_ws: ha_plan_bytes(nshard *
 * rcap). */
int ha_shard_frames_serve_push(float *table, int64_t rows, int64_t width, const uint32_t *keys_fixed, int nshard,
                               int64_t rcap, const float *values, void *plan_ws, ha_stream_t stream);

/* ---- HET embedding cache (replaces the hetu_cache plugin, src/hetu_cache) ----
 * An opaque device-resident cache of `limit` lines in front of a store (the "server": a table shard
 * plus one int64 version per row, ps-lite/include/ps/server/param.h:119-138).  Semantics of
 * CacheBase::_embeddingLookup / _embeddingUpdate / _embeddingUpdateWithPushKeys
 * (src/hetu_cache/src/cache.cc:60-107, 132-197, 248-335) with LRUCache eviction
 * (lru_cache.cc), LFUCache (lfu_cache.cc) or LFUOptCache (lfuopt_cache.cc) eviction.
 * policy: 0 = LRU, 1 = LFU, 2 = LFUOpt.  max_batch bounds the keys of one call.
 * key_kind: 0 = float32 ids (the *_raw entry points, cache.cc:49-58), 1 = uint64 keys.
 * dest / grads / keys are DEVICE pointers; all calls are asynchronous on `stream`. */
typedef struct ha_cache ha_cache;
ha_cache *ha_cache_create(int policy, int64_t limit, int64_t length,
                          int64_t width, int64_t max_batch);
void ha_cache_destroy(ha_cache *cache);
int ha_cache_set_bounds(ha_cache *cache, int64_t pull_bound, int64_t push_bound);
int ha_cache_set_bypass(ha_cache *cache, int bypass);
/* store rows [row_start, row_start+store_rows) of the global table live at `table`;
 * versions[store_rows] are the server-side row versions (zero-initialised by the caller). */
int ha_cache_bind_store(ha_cache *cache, float *table, int64_t *versions,
                        int64_t store_rows, int64_t row_start);
int ha_cache_lookup(ha_cache *cache, const void *keys, int key_kind, int64_t n,
                    float *dest, ha_stream_t stream);
/* The stable sort of a lookup's keys ONE BATCH EARLY, beside the calls of the current batch (the reference's loader hands
 * the ids over a batch ahead: python/hetu/dataloader.py:63-98).  `stream` forks into a stream of the cache's own here and
 * joins it in ha_cache_lookup_presorted(same keys, kind, n) -- anything else there is an error; the keys must not change in
 * between.  Batches beyond 36,864 keys are accepted and ignored (their lookup sorts by itself).  Results are those of
 * ha_cache_lookup, bit for bit (cache.cc:60-107). */
int ha_cache_sort_ahead(ha_cache *cache, const void *keys, int key_kind, int64_t n, ha_stream_t stream);
int ha_cache_lookup_presorted(ha_cache *cache, const void *keys, int key_kind, int64_t n,
                              float *dest, ha_stream_t stream);
/* The sorts of the next `count` (<= 16) lookups in ONE launch on `stream` -- no second stream, no event edge --, into a ring of
 * plan workspaces: for a caller that has its ids a block of batches early (the work-queue step's lookahead).  The
 * ha_cache_lookup_presorted calls that follow must name these key buffers in this order; 1 .. min(max_batch, 36,864) keys per
 * batch; the first call allocates the ring (not inside a stream capture).  Results: those of ha_cache_lookup. */
int ha_cache_sort_ahead_batch(ha_cache *cache, const void *const *keys, int key_kind, const int64_t *n, int count,
                              ha_stream_t stream);
/* The PLANNED flow (csrc/cache_block.hip; LRU, local store, limit >= max_batch): CacheBase::_embeddingLookup and
 * _embeddingUpdate of the same keys, batch after batch (src/hetu_cache/src/cache.cc:60-107, 132-197; the training loop of
 * python/hetu/cstable.py:38-56), with the BOOKKEEPING of a block of up to 16 batches done ahead: which lines a batch hits and
 * misses, the slots of the misses, the lines the policy evicts for them (LRUCache::insert lru_cache.cc:9-25; LFUCache
 * lfu_cache.cc:9-42 and LFUOptCache lfuopt_cache.cc:9-60: the least (use, arrival) of all lines, kept as a two-level minimum
 * by the bookkeeping; they require that every resident line was updated since its lookup when the planned flow takes over --
 * true after any lookup + update pair), update counters and the bounded push (cache.cc:159) follow from the ids alone.  ha_cache_plan_block enqueues, on `side`, the index plans of the
 * block's batches and ONE bookkeeping launch that leaves per batch the items of its two row launches; `main` = the stream of
 * those row launches (what is enqueued on it so far is ordered in front of the bookkeeping).  Then, per batch and in order,
 * ha_cache_lookup_planned (ONE launch: the staleness-bounded pull decided as the rows are read, rows to dest) and
 * ha_cache_update_planned (ONE launch: ordered accumulate, the pushed lines' store rows in the same pass, evicted dirty lines,
 * versions).  Results: those of ha_cache_lookup + ha_cache_update_same_keys call by call.  At most two blocks are outstanding
 * (plan block b + 1 when block b starts: its bookkeeping then runs beside block b's rows); the key buffers stay unchanged
 * until the bookkeeping has run; the call-by-call entry points (and ha_cache_set_bounds / _bypass) are refused while planned
 * calls are outstanding (ha_cache_plan_pending); ha_cache_perf after a planned call reports that call; ha_cache_state /
 * ha_cache_snapshot see the bookkeeping of every planned batch, i.e. are meaningful at the end of a block. */
int ha_cache_plan_block(ha_cache *cache, const void *const *keys, int key_kind, const int64_t *n, int count,
                        ha_stream_t side, ha_stream_t main);
int ha_cache_lookup_planned(ha_cache *cache, int64_t n, float *dest, ha_stream_t stream);
int ha_cache_update_planned(ha_cache *cache, int64_t n, const float *grads, ha_stream_t stream);
int ha_cache_plan_pending(ha_cache *cache);
/* `count` planned pairs by one call: ha_cache_lookup_planned(n[k], dests[k]) then ha_cache_update_planned(n[k], grads[k]) */
int ha_cache_run_planned_pairs(ha_cache *cache, int count, const int64_t *n, float *const *dests, const float *const *grads,
                               ha_stream_t stream);
int ha_cache_update(ha_cache *cache, const void *keys, int key_kind, int64_t n,
                    const float *grads, ha_stream_t stream);
/* ha_cache_update for the key batch of the immediately preceding ha_cache_lookup on this cache (the
 * training step's lookup / update pair): its index plan is still in the workspace and is reused. */
int ha_cache_update_same_keys(ha_cache *cache, int64_t n, const float *grads, ha_stream_t stream);
int ha_cache_update_with_push_keys(ha_cache *cache, const void *keys,
                                   int key_kind, int64_t n,
                                   const void *push_keys, int push_kind,
                                   int64_t n_push, const float *grads,
                                   ha_stream_t stream);
/* CacheBase::_embeddingPushPull (cache.cc:356-422): push the gradients of `push_keys`, then pull the
 * rows of `pull_keys` (the server pushes before it syncs) -- the ASP-prefetch call of
 * ParameterServerCommunicateOp (python/hetu/gpu_ops/ParameterServerCommunicate.py:37-40,68-72). */
int ha_cache_push_pull(ha_cache *cache, const void *pull_keys, int pull_kind,
                       int64_t n_pull, float *dest, const void *push_keys,
                       int push_kind, int64_t n_push, const float *grads,
                       ha_stream_t stream);
/* ---- the cache over a REMOTE store (rows owned by other ranks, or kept in host memory) ---------------
 * The reference's cache never touches the table: it talks to the servers through syncEmbedding /
 * pushEmbedding (src/hetu_cache/src/hetu_client.cc:6-39 -> ps-lite PSAgent.h:537-627 ->
 * PSFhandle_embedding.cc:5-64).  In remote mode the cache kernels do the same through two device
 * buffers the host side moves (herald_amd/cache.py: exchange of sharded.py, or staging from pinned
 * host memory):
 *   lookup  = ha_cache_lookup_begin   : plan + probe; REQUEST = (req_keys[u], req_versions[u]) for the
 *                                       n_unique keys of the batch, ascending (version -1 = no data);
 *             [the owner decides per key  pull = version == -1 || server_version - version > pull_bound
 *              (kSyncEmbedding, PSFhandle_embedding.cc:30-64) and answers INBOX: inbox_pull[u], and for
 *              pulled keys inbox_versions[u] and the row inbox_rows[inbox_idx[u], :]]
 *             ha_cache_lookup_finish  : touch / insert / evict bookkeeping, rows to `dest`, lines refreshed.
 *   update  = ha_cache_update* as before, except that the lines to push land in the OUTBOX instead of the
 *             store: entries [0, U) = the batch's unique keys in ascending order (out_keys[u] = key, or
 *             0xFFFFFFFF when that line is not pushed), entries [U, U+E) = pending evicted lines in eviction
 *             order; out_updates = the line's update count, out_rows = its gradient row.  The owner applies
 *             row += gradient, version += updates per entry, entries of one sender in list order
 *             (kPushEmbedding, PSFhandle_embedding.cc:5-28).  ha_cache_outbox_count = U + E.
 * embedding_push_pull over a remote store = update, exchange, lookup (the server pushes before it syncs,
 * PSFhandle_embedding.cc:66-79). */
typedef struct {
    uint32_t *req_keys;
    int64_t *req_versions;
    int32_t *inbox_pull, *inbox_idx;
    int64_t *inbox_versions;
    float *inbox_rows;
    uint32_t *out_keys;
    int32_t *out_updates;
    float *out_rows;
    int64_t out_capacity, max_batch;
} ha_cache_remote;
int ha_cache_set_remote(ha_cache *cache);
int ha_cache_remote_buffers(ha_cache *cache, ha_cache_remote *out);
int ha_cache_lookup_begin(ha_cache *cache, const void *keys, int key_kind, int64_t n,
                          int64_t *n_unique_host, ha_stream_t stream);
int ha_cache_lookup_finish(ha_cache *cache, int64_t n, float *dest, ha_stream_t stream);
int ha_cache_outbox_count(ha_cache *cache, int64_t *count_host, ha_stream_t stream);
/* Owners that need no host-side counts (a store on this device or in host memory: no exchange to size): the
 * REQUEST is always padded to n entries with key 0xFFFFFFFF (never pulled) -- pass n_unique_host = NULL and serve
 * n entries --, and after ha_cache_outbox_pad(entries) the updates mark OUTBOX entries [U + E, entries) as not
 * pushed, so outbox[0, entries) can be handed over without ha_cache_outbox_count.  entries must bound U + E: the
 * batch's n plus the n of every lookup since the last update (each evicts at most as many lines as it has keys). */
int ha_cache_outbox_pad(ha_cache *cache, int64_t entries);
/* embedding_push_pull (cache.cc:356-422) over a remote store: begin = touch of the pull keys + the whole
 * push phase (OUTBOX filled, REQUEST = the pull keys' versions after the push phase); the host pushes the
 * outbox and then syncs the request; finish = pull, rows to dest, insert of the pull misses, the push
 * phase's deferred version bump / zeroGrad. */
int ha_cache_push_pull_begin(ha_cache *cache, const void *pull_keys, int pull_kind, int64_t n_pull,
                             const void *push_keys, int push_kind, int64_t n_push,
                             const float *grads, int64_t *n_unique_pull_host, ha_stream_t stream);
int ha_cache_push_pull_finish(ha_cache *cache, float *dest, ha_stream_t stream);
/* Owner side of a remote lookup (kSyncEmbedding, PSFhandle_embedding.cc:30-64) for m requested shard-local
 * keys: pull[j] = versions[j] == -1 || server_versions[key] - versions[j] > bound; pulled rows are packed
 * in request order into rows_out, idx[j] = its row there, ver_out[j] = the server version; *count_dev
 * (device int64) = number of pulled rows.  scan_ws: reserved, may be NULL. */
int ha_store_serve_sync(const float *table, const int64_t *server_versions, int64_t rows, int64_t width,
                        const uint32_t *keys, const int64_t *versions, int64_t m, int64_t bound,
                        int32_t *pull, int32_t *idx, int64_t *ver_out, float *rows_out,
                        int64_t *count_dev, void *scan_ws, ha_stream_t stream);
/* Owner side of the version half of a push: server_versions[keys[j]] += updates[j] (keys >= rows skipped). */
int ha_store_add_versions(int64_t *server_versions, int64_t rows, const uint32_t *keys,
                          const int32_t *updates, int64_t m, ha_stream_t stream);
/* The same server side for a push list whose live keys (< rows; 0xFFFFFFFF = not pushed) are pairwise DISTINCT -- what
 * ha_cache_update_same_keys' two-launch path leaves in the outbox (the batch's unique keys and the victims of one lookup):
 * one launch, table[key,:] += grad_rows[j,:] and server_versions[key] += updates[j], no sort.  The caller vouches for the
 * distinctness (herald_amd.cache asks ha_cache_fused_updates); lists with repeated keys need ha_shard_serve_push's order. */
int ha_store_push_distinct(float *table, int64_t *server_versions, int64_t rows, int64_t width, const uint32_t *keys,
                           const int32_t *updates, const float *grad_rows, int64_t m, ha_stream_t stream);
/* *acc += number of keys below `rows` (device-side traffic accounting of a store) */
int ha_store_count_valid(const uint32_t *keys, int64_t m, int64_t rows, int64_t *acc, ha_stream_t stream);
/* Synchronising inspectors.  perf out[8] = {type(0 pull / 1 push), num_all, num_unique, num_miss,
 * num_transfered, num_evict, is_full, size} of the last call (the perf dict of cache.cc:89-106).
 * state out[8] = {size, pending evictions, free slots, log head, log tail, clock, slots, log cap}. */
int ha_cache_perf(ha_cache *cache, int64_t *out_host, ha_stream_t stream);
int ha_cache_state(ha_cache *cache, int64_t *out_host, ha_stream_t stream);
/* Diagnostics: out_host[16] = the GPU's 100 MHz clock at the phase boundaries of the last lookup's bookkeeping
 * ([0..4] the bookkeeping workgroup, [8..12] the insert / eviction workgroup beside the row copies). */
int ha_cache_phase_times(ha_cache *cache, uint64_t *out_host, ha_stream_t stream);
/* Stage times of the last lookup / update in milliseconds, the fields of the reference's perf dict (`time, sort_time,
 * lookup_time, prepare_time, transfer_time, copy_time, insert_time / cleanup_time`, src/hetu_cache/src/cache.cc:99-105,
 * 189-194; dumped by examples/ctr/run_hetu.py:508-515), measured with HIP events between the call's launches once
 * ha_cache_set_timing(cache, 1) was called (off by default).  out_ms[6] = {whole call, sort, lookup (+ prepare), copy (+
 * insert), transfer, the rest}; -1 = a stage the call did not pass. */
int ha_cache_set_timing(ha_cache *cache, int on);
int ha_cache_stage_times(ha_cache *cache, double *out_ms);
/* Resident lines, unordered, into device arrays of capacity cap; *count_dev (zeroed by the
 * caller) receives the number of resident lines; slots[] index the rows of ha_cache_data/grad. */
int ha_cache_snapshot(ha_cache *cache, int64_t cap, uint32_t *keys,
                      int64_t *version, int32_t *updates, uint64_t *stamp,
                      int32_t *slots, uint64_t *count_dev, ha_stream_t stream);
/* Debug hook behind hetu_cache's insert(EmbeddingPT) (src/hetu_cache/src/python_api.cc:58, lru_cache.cc:9-25):
 * the RESIDENT line of `key` takes `version` and the `width` floats at data_dev; a key that is not resident
 * is left alone (the caller brings the line in with a one-key lookup first). */
int ha_cache_set_line(ha_cache *cache, int64_t key, int64_t version, const float *data_dev, ha_stream_t stream);
float *ha_cache_data(ha_cache *cache);
float *ha_cache_grad(ha_cache *cache);
int64_t ha_cache_limit(ha_cache *cache);
int64_t ha_cache_width(ha_cache *cache);
/* Updates that took the two-launch path of ha_cache_update_same_keys (LRU, local store, the keys of the preceding
 * lookup, limit >= batch: accumulate + one launch for touch / push / commit) since the cache was created.
 * HA_CACHE_FUSED=0 in the environment keeps every update on the general path. */
int64_t ha_cache_fused_updates(ha_cache *cache);

/* ---- laia embedding scheduler (replaces the laia_cache plugin, laia/) --------
 * LaiaScheduler::get_dist + the snapshot update of launch() (laia/src/laia_scheduler.cc:115-271) for
 * one global batch of nrank*mini_bs samples: probing / plan extraction on the GPU, the greedy
 * assignment and the MiniLRUCache bookkeeping (laia/include/mini_lru_cache.h) on the calling host
 * thread.  samples_host[num_sample*num_table] are the embedding keys (all < key_limit,
 * nrank*key_limit < 2^32).  ha_laia_next writes dist_out[nrank*mini_bs] (global sample indices per
 * worker) and the workers' sorted communication plans concatenated into plan_out with
 * plan_off[nrank+1].  All pointers are HOST pointers; the call synchronises its private stream. */
typedef struct ha_laia ha_laia;
ha_laia *ha_laia_create(const uint64_t *samples_host, int64_t num_sample,
                        int64_t num_table, int64_t nrank, int64_t cache_size,
                        int64_t key_limit, int64_t max_batch);
void ha_laia_destroy(ha_laia *sched);
int ha_laia_next(ha_laia *sched, int64_t batch_id, int64_t mini_bs,
                 int64_t *dist_out, uint64_t *plan_out, int64_t plan_cap_elems,
                 int64_t *plan_off);
/* One batch ahead: the caller announces the batch of its NEXT ha_laia_next* call (the reference's launch() walks them in
 * order, laia/src/laia_scheduler.cc:115-169); the call that follows the hint enqueues the announced batch before it returns
 * its own results.  The announced call must follow; scheduler state read in between is one batch ahead.  Device-resident
 * state only (ignored otherwise); next_batch_id < 0 withdraws an unused hint. */
int ha_laia_hint_next(ha_laia *laia, int64_t next_batch_id);
/* ha_laia_next for a caller that wants ONE worker's plan (what LaiaScheduler::launch queues for its own rank,
 * laia/src/laia_scheduler.cc:140-168): plan_out holds that plan, plan_off[w] = 0 for w <= rank and its length behind;
 * dist_out as ha_laia_next.  With the scheduler state on the device only that worker's plan rows cross PCIe. */
int ha_laia_next_for_rank(ha_laia *h, int64_t batch_id, int64_t mini_bs, int64_t rank, int64_t *dist_out,
                          uint64_t *plan_out, int64_t plan_cap_elems, int64_t *plan_off);
int64_t ha_laia_snapshot_keys(ha_laia *sched, int64_t worker, int32_t *out,
                              int64_t cap);

/* TopkScheduler::get_dist + the snapshot update of its launch() for one global batch
 * (laia/src/topk_scheduler.cc:362-502, 319-345): only the tables table_order[0..top_k) are scored, the
 * batch and every worker's quota are cut into num_threads slices assigned independently, a sample is
 * offered to the workers starting at the one that first reached its top score, and plan[w] = keys of w's
 * own samples that w's snapshot holds valid.  Outputs as ha_laia_next.  num_threads must divide the
 * work so that no slice holds more samples than nrank x its quota (the reference writes out of bounds
 * there). */
int ha_laia_next_topk(ha_laia *sched, int64_t batch_id, int64_t mini_bs,
                      const int32_t *table_order, int64_t top_k, int64_t num_threads,
                      int64_t *dist_out, uint64_t *plan_out, int64_t plan_cap_elems,
                      int64_t *plan_off);
/* out[4*nrank] = miss_pull, miss_push, update_pull, update_push per worker, accumulated over the
 * batches scheduled so far (TopkScheduler::report_cache_perf, topk_scheduler.cc:504-527). */
int ha_laia_counters(ha_laia *sched, int64_t *out);
/* Wall time of the scheduler per phase, summed since ha_laia_create (scheduler us per global
 * batch): out[4] = {calls of ha_laia_next*, whole calls us, host greedy assignment us, host snapshot
 * (MiniLRU) bookkeeping us}; the remainder of the total is GPU kernels, transfers and waits. */
int ha_laia_timing(ha_laia *h, double *out);
/* device-resident mode: out[4] = {calls, us enqueueing the next batch's launches, us waiting for a batch's results, us copying
 * dist and plan out of the pinned mirror}, summed since creation */
int ha_laia_timing_device(ha_laia *h, double *out);
/* 1 = the scheduler state (MiniLRU snapshots as stamp logs, assignment, sorted-unique plan / touched rows) lives on the
 * device -- LaiaScheduler and TopkScheduler alike, whenever cache_size >= global batch x tables (decided at the first
 * batch; HA_LAIA_HOST=1 keeps it on the host) --, 0 = host snapshots, -1 = no batch scheduled yet. */
int ha_laia_on_device(ha_laia *h);

/* Local-shared plan distribution (laia/include/share_mem.h:40-193, ring_buffer.h:13-125): a
 * single-producer / single-consumer ring of uint64 words in POSIX shared memory, message-framed.
 * The scheduler of local rank 0 creates "laia_cache_<i>" for every local worker i and sends it its
 * [plan, dist] stream; worker i opens its ring and receives.  send: 1 sent / 0 no room yet / -1 never
 * fits.  recv: message length / -1 empty / -2 `out` too small (*needed = length). */
typedef struct ha_shm_ring ha_shm_ring;
ha_shm_ring *ha_shm_ring_open(const char *name, int create, int64_t capacity_words);
void ha_shm_ring_close(ha_shm_ring *ring);
int ha_shm_ring_send(ha_shm_ring *ring, const uint64_t *words, int64_t n);
int64_t ha_shm_ring_recv(ha_shm_ring *ring, uint64_t *out, int64_t cap_words, int64_t *needed);
int64_t ha_shm_ring_pending_words(ha_shm_ring *ring);

/* Development aid: ha_sgd_apply with per-wave time stamps, dbg[4*n] u64 =
 * {s_memrealtime start, end (10 ns ticks), role/len, shader cycles} per sorted position. */
int ha_debug_apply_timeline(float *table, int64_t rows, int64_t width,
                            const void *plan_ws, int64_t n, const float *grads,
                            float lr, unsigned long long *dbg,
                            ha_stream_t stream);

/* One-call convenience used by the reference-named SGDOptimizerSparseUpdate:
 * plan + apply using an internal per-stream workspace. */
int ha_sgd_sparse_update_f32ids(float *table, int64_t rows, int64_t width,
                                const float *ids, int64_t n,
                                const float *grads, float lr,
                                ha_stream_t stream);

/* ---- in-node parameter-server engine behind the libps names (include/herald_ps.h) --------------------
 * ps-lite/src/python_binding.cc:6-151 -> Worker -> PSAgent (PSAgent.h:124-237) -> servers
 * (PSFHandle.h:101-164, 401-439).  Every process owns the AveragePartitioner row range of its rank
 * (partitioner.h:46-57) of every tensor, in its GPU's HBM.  Index / value DLArrays may be device or host
 * arrays.  Calls are asynchronous on the tensor's stream; ha_ps_wait joins it.  With nrank > 1 the sparse
 * exchange is provided by the host side as a backend (herald_amd/ps.py registers the RCCL all-to-all of
 * herald_amd/sharded.py); without one everything is served locally. */
typedef struct {
    /* ids_dev: float32 ids (global row numbers); out_dev / vals_dev: [n, width]; stream: the tensor's */
    int (*sparse_pull)(int node, const float *ids_dev, int64_t n, float *out_dev, void *stream);
    int (*sparse_push)(int node, const float *ids_dev, int64_t n, const float *vals_dev, void *stream);
    int (*barrier)(void);
} ha_ps_backend;
typedef struct {
    float *table;          /* this rank's rows [row_start, row_start + rows_local) */
    int64_t len, width, row_start, rows_local;
    void *stream;
} ha_ps_tensor_info;
int ha_ps_configure(int rank, int nrank);
int ha_ps_set_backend(const ha_ps_backend *backend);
int ha_ps_rank(void);
int ha_ps_nrank(void);
/* init_type: 0 Constant(a), 1 Uniform(a, b), 2 Normal(mean a, stddev b), 3 TruncatedNormal (psf/misc.h:7-12);
 * values are a function of (seed, element index) only: the same table whatever the number of ranks */
int ha_ps_init_tensor(int node, int ptype, int64_t len, int64_t width, int init_type,
                      double a, double b, uint64_t seed);
/* serve an existing device shard (rows of this rank's range) instead of allocating one */
int ha_ps_attach_tensor(int node, float *table_dev, int64_t len, int64_t width);
int ha_ps_tensor(int node, ha_ps_tensor_info *out);
int ha_ps_sparse_pull(int node, const DLArray *index, DLArray *value);
int ha_ps_sparse_push(int node, const DLArray *index, const DLArray *value);
int ha_ps_dense_pull(int node, DLArray *arr);
int ha_ps_wait(int node);
int ha_ps_barrier(void);
int ha_ps_clear(int node);
int ha_ps_save(int node, const char *address);   /* <address>/<node>_<rank>.dat, raw fp32 rows */
int ha_ps_load(int node, const char *address);

#ifdef __cplusplus
}
#endif
#endif /* HERALD_AMD_H_ */
