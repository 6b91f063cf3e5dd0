/*
 * gnnagg.h -- C-ABI of libgnnagg.so, the MI355X-native neighbor-aggregation library.
 *
 * This is the drop-in boundary for the hot path of xxcclong/GNN-Computing
 * (include/aggregator.h, aggr_gcn.h, aggr_gat.h, spmm.h, graph_schedule.h, src/data.cu).
 * Plain pointers and sizes only; every `d_*` / "device" pointer is HIP device memory owned by
 * the caller, every `h_*` / "host" pointer is host memory.  No torch / C++ types cross it.
 *
 * Section A mirrors, name for name, the flat API the reference's own PyTorch binding is written
 * against (reference Figure7/kernel.cpp:15-35, defined in Figure7/kernel_generated.cu:15-74);
 * a reference-side binding only has to link this library instead of compiling the CUDA headers.
 * Section B is the same functionality with status codes, streams, reductions and heads.
 * Section C is the host graph preparation (reference src/data.cu, include/graph_schedule.h).
 * Section D is the 1-D row-partition / halo-exchange support (no reference counterpart:
 * the reference asserts GPUNUM == 1, Figure9/main.cu:19).
 *
 * Error model: Section A keeps the reference's abort-on-error behaviour (include/util.h:82-104:
 * message with the failing call, then exit(1)) unless gnnagg_set_abort_on_error(0) was called, in
 * which case the error is recorded and the call returns (0 for the *_init functions).
 * Sections B-D return GNNAGG_OK or an error code; gnnagg_last_error() gives the text.
 * All `run` entry points are asynchronous on the handle's stream (reference: run() never
 * synchronises, aggr_gcn.h:396,407); schedule / create are synchronous.
 */
#ifndef GNNAGG_H
#define GNNAGG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GNNAGG_OK 0
#define GNNAGG_ERR_ARG 1    /* bad argument / contract violation (reference: assert) */
#define GNNAGG_ERR_HIP 2    /* HIP runtime failure (reference: checkCudaErrors) */
#define GNNAGG_ERR_STATE 3  /* e.g. scheduled run without a schedule (reference: assert aggr_gcn.h:392) */
#define GNNAGG_ERR_IO 4     /* graph files missing / malformed (reference: assert(fexist), data.cu:40) */

/* enum Schedule, reference include/graph_schedule.h:8-14 (same values) */
#define GNNAGG_SCHED_LOCALITY 0
#define GNNAGG_SCHED_NEIGHBOR_GROUPING 1
#define GNNAGG_SCHED_LOCALITY_NEIGHBOR_GROUPING 2
#define GNNAGG_SCHED_NOP 3

/* reduction over a row's neighbors */
#define GNNAGG_REDUCE_SUM 0  /* reference aggr_gcn.h:13-35; val == NULL means weight 1 */
#define GNNAGG_REDUCE_MEAN 1 /* sum then / deg   (reference: caller passes val = 1/deg) */
#define GNNAGG_REDUCE_MAX 2  /* max_e val*x, 0 for an empty row (no reference kernel) */

/* run modes */
#define GNNAGG_MODE_ROWS 0      /* `scheduled = 0`: one work item per CSR row, sequential FMA chain in CSR order */
#define GNNAGG_MODE_SCHEDULED 1 /* `scheduled = 1`: work items = groups of the last schedule() call */
#define GNNAGG_MODE_BALANCED 2  /* library-chosen chunking of long rows (gnnagg_schedule_balanced) */

typedef int64_t gnnagg_handle; /* opaque; same width as the reference's `int64_t at` (kernel.cpp:15) */

const char *gnnagg_last_error(void);
int gnnagg_version(void);
void gnnagg_set_abort_on_error(int on);

/* ---------------------------------------------------------------------------------------------
 * A. Flat API -- one-to-one with reference Figure7/kernel.cpp:15-35
 * ------------------------------------------------------------------------------------------- */
/* kernel.cpp:15 / kernel_generated.cu:15-19.  ptr[num_v+1], idx[num_e], val[num_e] on device,
 * borrowed for the life of the handle (the reference's torch path never frees them either). */
int64_t GCN_init_impl(int *ptr, int *idx, float *val, int num_v, int num_e);
/* kernel.cpp:17 / kernel_generated.cu:21-24 (Aggregator_GCN::updateval, aggr_gcn.h:540-544) */
void GCN_update_val_impl(int64_t at, float *val);
/* kernel.cpp:19 / kernel_generated.cu:26-32 (run_with_feat, aggr_gcn.h:411-444).  `blocksize` is a
 * CUDA block-geometry hint in the reference; accepted and ignored. */
void GCN_run_impl(int64_t at, float *feat, float *out_feat, int blocksize, int scheduled, int featlen);
/* kernel.cpp:21 / kernel_generated.cu:34-39: schedule(neighbor_grouping, arr), arr[0] = NG */
void GCN_schedule_impl(int64_t at, int *arr);
/* kernel.cpp:23 / kernel_generated.cu:41-45 */
int64_t GAT_init_impl(int *ptr, int *idx, int num_v, int num_e);
/* kernel.cpp:25 / kernel_generated.cu:47-50 (Aggregator_GAT::run_with_feat, aggr_gat.h:355-394);
 * att is [V,2]; leaky slope 0.2 (aggr_gat.h:347) */
void GAT_run_impl(int64_t at, float *feat, float *att, float *out_feat, int blocksize, int scheduled, int featlen);
/* kernel.cpp:27-31 / kernel_generated.cu:52-65 (aggr_gat.h:402-425) */
void GAT_run_u_add_v_impl(int64_t at, float *att, float *outval, int blocksize);
void GAT_run_add_to_center_impl(int64_t at, float *inval, float *outatt, int blocksize);
void GAT_run_div_each_impl(int64_t at, float *inatt, float *inoutval, int blocksize);
/* kernel.cpp:35 / kernel_generated.cu:69-74 */
void GAT_schedule_impl(int64_t at, int *arr);

/* ---------------------------------------------------------------------------------------------
 * B. Status-returning API (streams, reductions, heads, lifetime)
 * ------------------------------------------------------------------------------------------- */
/* Aggregator_GCN ctor, aggr_gcn.h:365-374.  d_val may be NULL (implicit weight 1). */
int gnnagg_gcn_create(const int *d_ptr, const int *d_idx, const float *d_val, int num_v, int num_e,
                      gnnagg_handle *out);
/* Aggregator_GAT ctor, aggr_gat.h:302-313 */
int gnnagg_gat_create(const int *d_ptr, const int *d_idx, int num_v, int num_e, gnnagg_handle *out);
/* Frees what the handle allocated (schedules, scratch).  Never frees caller pointers. */
int gnnagg_destroy(gnnagg_handle h);
/* hipStream_t as void*; NULL = default stream.  Work of later calls is enqueued there, and whatever the library reads of the caller's
 * arrays on the HOST (the CSR, when a schedule or plan is built) is copied on that stream and waited for -- i.e. ordered behind the
 * caller's work on it.  Set it before schedule() / the first run when the arrays were produced on a non-null stream. */
int gnnagg_set_stream(gnnagg_handle h, void *hip_stream);
/* Per-handle knobs.  Every knob is an option; the four a C++ driver linked against the class shim cannot reach through code also
 * read an environment variable (in brackets) when the handle is made.  INTEGRATION.md section 5 is the table of all of them.
 *   "partitions" [GNNAGG_PARTITIONS]      -1 the library decides (average degree >= 96), 0 never, N > 0: N source ranges for the balanced mode
 *   "tile_width"                          floats per column tile of the 2-D blocked balanced mode: 32 / 64 / 128 / 256 (64)
 *   "slice_kb"                            target size of the X slice an XCD's L2 holds (4096)
 *   "scratch_limit_mb"                    > 0: cap on the scratch (partial rows + tiled image of X) the blocked order may take;
 *                                         a handle that would need more -- or more than half of the free device memory, or
 *                                         whose allocation fails -- moves to the chunked plan for good
 *   "fast_rows" [GNNAGG_FAST_ROWS]        1: GNNAGG_MODE_ROWS (`scheduled = 0`) runs the balanced order -- results within the
 *                                         1e-5 bound instead of bit-exact CSR-order chains; 0: canonical order.  Default 0 for
 *                                         handles made through this section, 1 for the reference-facing surfaces (see
 *                                         "reference_defaults"); the environment variable overrides both
 *   "reference_defaults"                  1: what GCN_init_impl / GAT_init_impl, the C++ class shim (include/compat) and the pybind-
 *                                         named Python functions apply to the handles they make: fast_rows = 1 unless
 *                                         GNNAGG_FAST_ROWS says otherwise.  A reference driver's run(vin, vout, B, 0) is then as
 *                                         fast as its scheduled run; aggr_gcn's own result differs from it by association only
 *   "fast_scheduled" [GNNAGG_FAST_SCHEDULED]  1 (default): GNNAGG_MODE_SCHEDULED (`scheduled = 1`) runs the balanced order.  The
 *                                         reference's scheduled kernels add group partials with atomicAdd (aggr_gcn.h:112,
 *                                         aggr_gat.h:196-203): every association is one of its legal results.  A schedule must
 *                                         still have been made; num_target / get_schedule / mode_params of GNNAGG_MODE_SCHEDULED
 *                                         keep describing the user's groups, the order that RUNS is the one GNNAGG_MODE_BALANCED's
 *                                         queries describe; GAT calls that ask for newval keep the scheduled order -- and so does a
 *                                         locality schedule that DROPPED edges (total_num_v below the largest column id + 1,
 *                                         graph_schedule.h:23-44): the balanced order covers every edge.  0: the user's groups,
 *                                         folded in the restated order (bit-exact against the oracle)
 *   "aux_stream" [GNNAGG_AUX_STREAM]      0: GNNAGG_MODE_ROWS runs its hub rows on the handle's stream, before the short rows, instead of
 *                                         beside them on an auxiliary stream (slower by the hub rows' duration, but the process keeps
 *                                         a single queue); 1 (default)
 *   "rows_blocked"                        1 (default): GNNAGG_MODE_ROWS runs its canonical chains on the 2-D blocked order where the
 *                                         graph allows it (gnnagg_rows_blocked_ranges); 0: always the row kernels.  Same bits either way
 *   "rows_medium_edges"                   GNNAGG_MODE_ROWS on the row kernels: rows above this many edges, up to the hub threshold
 *                                         max(1024, 16 x mean degree), run on 128-thread workgroups (one gather wavefront + the chain)
 *                                         instead of one lane group each.  0 (default): max(128, E / 4500), an empty class from 4.6 M
 *                                         edges on; -1: no such class.  Same bits whatever the value
 *   "rows_hub_tile"                       GNNAGG_MODE_ROWS, hub rows (GCN flavours): 32- or 64-column tiles per workgroup of the long-row
 *                                         kernel; 0 (default): 64 where the launch has more (row, tile) items than twice the CUs.  Same bits
 *   "rows_hub_edges"                      chained rows mode: rows with a (row, range) sub-row above this many edges leave the chained
 *                                         launches for the long-row kernel (0: the library's rule)
 * Twelve options.  [GNNAGG_XCD_REMAP] 0 / 1 / 2 (workgroup -> XCD mapping: identity / equal-count / work-balanced ranges, default 2) is an
 * environment-only measurement switch.  libgnnagg_extras.so (-DGNNAGG_EXTRAS, Section E) additionally knows "partition_min_degree",
 * the older forms "retile", "tiled", "spans", "inkernel_combine", "host_plan" and [GNNAGG_PLAN]: second-tier A / B material, constants in
 * the shipped library.
 * Options that change the library-chosen order drop it; it is rebuilt on the next use. */
int gnnagg_set_option(gnnagg_handle h, const char *name, int value);
/* What the library-chosen blocked order cost to build and holds (the reference prints its schedule time, graph_schedule.h:125-127):
 * wall seconds of the last construction of the balanced mode's 2-D blocked order (0: the handle runs the chunked plan, built in O(V))
 * and of the rows mode's chain plan; device bytes of the plans' arrays; device bytes of the scratch the runs so far have reserved
 * (partial rows, tiled images of X / Y, compact attention terms).  Any output pointer may be NULL. */
int gnnagg_plan_info(gnnagg_handle h, double *plan_seconds, double *rows_plan_seconds, long long *plan_bytes, long long *scratch_bytes);
/* Aggregator_GCN::updateval, aggr_gcn.h:540-544: re-aliases the edge values (borrowed; read at run time). */
int gnnagg_update_val(gnnagg_handle h, const float *d_val);

/* Per-row degrees for aggregations that are computed in TWO passes over disjoint edge sets of the same rows (the row-partitioned
 * step of section D: local-source edges while the halo exchange is in flight, halo-source edges after it).  d_row_aux[num_v]
 * (int32, borrowed, read at run time; NULL switches it off) changes GNNAGG_MODE_BALANCED runs of this handle as follows:
 *   GNNAGG_REDUCE_MEAN  divides by d_row_aux[row] -- the row's degree in the WHOLE graph -- instead of the edges this handle holds
 *                       (the reference's mean is val = 1/deg per edge, Figure7/our.py: every term is x/deg, so the two passes add);
 *                       GNNAGG_FLAG_ACCUMULATE then adds the pass's quotient to what the row holds
 *   GNNAGG_REDUCE_MAX   with GNNAGG_FLAG_ACCUMULATE: y = max(y, result) where d_row_aux[row] > 0 edges were already folded into y
 *                       by the earlier pass, y = result where none were (y then holds the empty row's 0, not a maximum)
 * GNNAGG_REDUCE_SUM is not affected.  Rows this handle has no edges for are left untouched by an accumulating run. */
int gnnagg_set_row_aux(gnnagg_handle h, const int *d_row_aux);

/* Aggregator::schedule, aggregator.h:67-99 / Aggregator_GCN::schedule aggr_gcn.h:501-538.
 * kind = GNNAGG_SCHED_*; param[0] = NG or par_num, param[1] = NG for the combined schedule.
 * total_num_v: the reference reads the global `n` (aggregator.h:79); pass num_v for a full graph. */
int gnnagg_schedule(gnnagg_handle h, int kind, const int *param, int total_num_v);
/* Library-chosen chunking of long rows into work items of <= chunk edges (0 = choose from the
 * degree distribution).  Used by GNNAGG_MODE_BALANCED. */
int gnnagg_schedule_balanced(gnnagg_handle h, int chunk);
/* Summation order of GNNAGG_MODE_BALANCED: rows are cut into chunks of *chunk edges (partial FMA chains
 * from 0, like the reference's neighbor groups); seg_chunks > 0 means the chunk partials of a row are folded
 * in ascending order inside segments of seg_chunks chunks and the segment sums are then added in ascending
 * order; 0 means one flat ascending fold.  (Rows of at most seg_chunks chunks are identical either way.) */
int gnnagg_balanced_params(gnnagg_handle h, int *chunk, int *seg_chunks);
/* The same for any mode.  GNNAGG_MODE_SCHEDULED with a neighbor-grouping schedule: *chunk = NG and *seg_chunks = 16
 * when the plan kernel runs it (the reference adds the group partials with fp32 atomics in arbitrary order,
 * aggr_gcn.h:112, so every fixed order is one of its outcomes up to association), 0 (flat ascending fold) when NG is
 * so small that the one-item-per-lane-group kernel is used, and always 0 for the locality schedules.
 * GNNAGG_MODE_ROWS: one chain per row (*chunk = INT_MAX, *seg_chunks = 0). */
int gnnagg_mode_params(gnnagg_handle h, int mode, int *chunk, int *seg_chunks);
/* Source partitions of the balanced mode: 0 for the chunked order reported by gnnagg_balanced_params; P > 0 when the library
 * chose the source-partitioned (2-D blocked) order for a high-degree graph (avg degree >= 96): the groups are those of
 * gnnagg_locality_schedule(par_num = P, neighbor_num = chunk, total = *total_cols) -- partition-major, row-minor, CSR order
 * inside a sub-row -- folded flat in ascending group order per row; gnnagg_get_schedule(h, GNNAGG_MODE_BALANCED, ...) returns
 * them.  *total_cols (may be NULL) = largest neighbor id + 1: the column count the ranges are cut from (the CSR need not be
 * square). */
int gnnagg_balanced_partitions(gnnagg_handle h, int *partitions, int *total_cols);
/* GNNAGG_MODE_ROWS of a GCN handle: *ranges = the number of source ranges when the canonical chains run on the 2-D blocked order
 * (option "rows_blocked", default 1: graphs of average degree >= "partition_min_degree" whose rows list their neighbors in ascending
 * order -- range after range is then the CSR order, and every (row, column) stays the reference's one sequential chain,
 * aggr_gcn.h:13-35, with the gathers served by the L2; sum / mean, feature widths above 32), 0 when the row kernels run. */
int gnnagg_rows_blocked_ranges(gnnagg_handle h, int *ranges);
/* Aggregator::num_target (aggregator.h:126), and the scheduled arrays copied to host buffers
 * (any may be NULL): ptr_s[num_target+1], idx_s[ptr_s[num_target]], target[num_target], val_s. */
int gnnagg_num_target(gnnagg_handle h, int mode, int *out);
int gnnagg_get_schedule(gnnagg_handle h, int mode, int *h_ptr_s, int *h_idx_s, int *h_target, float *h_val_s);

/* Aggregator_GCN::run / run_with_feat, aggr_gcn.h:379-444.  x,y are [V,feat] row-major fp32 on
 * device.  mode = GNNAGG_MODE_*, reduce = GNNAGG_REDUCE_*.  y is fully overwritten. */
int gnnagg_gcn_run(gnnagg_handle h, const float *d_x, float *d_y, int feat, int mode, int reduce);
/* Same with flags.  GNNAGG_FLAG_ACCUMULATE (balanced mode, sum): y += A.x -- the row's result is computed as
 * usual and added to the value already in y with one fp32 add; rows without edges are left untouched.  Used by the
 * row-partitioned path to add the halo-column part after the local-column part. */
#define GNNAGG_FLAG_ACCUMULATE 1
/* GNNAGG_FLAG_RELU (any mode / reduce): y = max(result, 0) -- the activation that follows the aggregation in the
 * reference's 3-layer model (Figure7/our.py:176, F.relu on the output of gcn_run), applied to the finished row by the
 * producing kernel instead of one more pass over y.  With GNNAGG_FLAG_ACCUMULATE: y = max(y + A.x, 0), rows without
 * edges included. */
#define GNNAGG_FLAG_RELU 2
int gnnagg_gcn_run_ex(gnnagg_handle h, const float *d_x, float *d_y, int feat, int mode, int reduce, int flags);
/* Measurement aid (no reference counterpart): the gather ceiling of gnnagg_gcn_run(h, d_x, ., feat, mode, SUM).  Launches
 * the same kernel over the same work items with the same descriptor / neighbor-id / edge-value loads and the same
 * feature-row gathers (same addresses, same batching), but consumes the data with integer XORs instead of the
 * dependent FMA chains and stores nothing.  Its duration is what the memory system needs to deliver this launch's
 * gathers; bench.py reports roofline.frac = probe time / kernel time.  GNNAGG_MODE_BALANCED (or a neighbor-grouping
 * schedule that runs on the plan kernel); asynchronous on the handle's stream. */
int gnnagg_gcn_probe_gather(gnnagg_handle h, const float *d_x, int feat, int mode);
/* The same for gnnagg_gat_run on the 2-D blocked balanced order (k_gat_span): neighbor ids, compact attention terms and
 * tile-row gathers as in the real run, no exp, no chain, no store.  GNNAGG_ERR_ARG on the other GAT paths. */
int gnnagg_gat_probe_gather(gnnagg_handle h, const float *d_x, const float *d_att, int feat, int heads, int mode);
/* Measurement aid (no reference counterpart): the rate the memory system offers to ROW GATHERS in these kernels' own access shape,
 * independent of any graph.  One launch; lane groups of L lanes (L = the power of two >= seg_bytes / 16, 8 .. 64; 256 / L groups per
 * workgroup) each take `ids_per_group` consecutive entries of d_ids (coalesced id loads, 8 gathers in flight) and read seg_bytes at
 * d_rows + id * pitch_bytes; the data is XOR-consumed and never stored.  Where the gathered rows live -- one XCD's L2, the Infinity
 * Cache, HBM -- is the caller's choice of ids (workgroup b runs on XCD b % 8).  seg_bytes and pitch_bytes multiples of 16,
 * ids_per_group a multiple of L, n_ids a multiple of ids_per_group x 256 / L.  Asynchronous on hip_stream.  bench.py times this
 * launch to quote each roofline fraction against a ceiling that bounds the bytes it divides (gather-model bytes are cache-served). */
int gnnagg_probe_row_gather(const void *d_rows, long long pitch_bytes, int seg_bytes, const int *d_ids, long long n_ids, int ids_per_group,
                            void *hip_stream);
/* Aggregator_GCN::run_clock, aggr_gcn.h:462-489 (Figure 8 load-balance study).  Runs the one-item-per-lane-group
 * kernel of mode rows (the reference's aggr_gcn_clock) or scheduled (aggr_gcn_target_clock) with per-workgroup
 * stamps: d_timer[3b] = start, [3b+1] = end (ticks of the constant wall clock, gnnagg_wall_clock_hz), [3b+2] = CU id.
 * Call with d_timer == NULL to get *num_blocks (the timer needs 3 * num_blocks entries; pass the run's d_x / d_y to the
 * query when they may be less than 16-byte aligned -- the lane geometry, and with it the grid, follows their alignment).
 * With d_timer != NULL, *num_blocks is in/out: in = the workgroups d_timer has room for (GNNAGG_ERR_ARG when the launch
 * needs more), out = the workgroups launched. */
int gnnagg_gcn_run_clock(gnnagg_handle h, const float *d_x, float *d_y, int feat, int mode, unsigned long long *d_timer,
                         int *num_blocks, int *waves_per_cu);
long long gnnagg_wall_clock_hz(void);
/* Aggregator_GCN::runEdgeWise, aggr_gcn.h:446-460 (edge-parallel atomics; any feat). */
int gnnagg_gcn_run_edgewise(gnnagg_handle h, const float *d_x, float *d_y, int feat);
/* matmul_NN, include/dense.h:4-23: c[m,n] = a[m,k] . b[k,n], row-major fp32 (the dense combine after an
 * aggregation).  f32 MFMA, accumulation in ascending k. */
int gnnagg_matmul_nn(const float *d_a, const float *d_b, float *d_c, int m, int n, int k, void *hip_stream);
/* Aggregator_GCN::run_with_nn, aggr_gcn.h:491-499 (kernel aggr_gcn_nn :304-359): y = A.x, then
 * transformed[V,feat_out] = y . weight[feat_in,feat_out].  Both outputs are fully overwritten (the
 * reference accumulates into whatever they held). */
int gnnagg_gcn_run_with_nn(gnnagg_handle h, const float *d_x, float *d_y, const float *d_weight, float *d_transformed,
                           int feat_in, int feat_out, int mode);
/* Input check the reference does not have (an out-of-range neighbor id is a silent out-of-bounds gather there):
 * *bad_rows = rows with ptr[r] > ptr[r+1], *bad_indices = neighbor ids outside [0, num_cols) (num_cols <= 0: num_v).
 * Synchronises the handle's stream. */
int gnnagg_check_csr(gnnagg_handle h, int num_cols, int *bad_rows, int *bad_indices);
/* Aggregator::csr2edgelist, aggregator.h:115-122: d_edgelist[2E] = (src, dst) pairs */
int gnnagg_csr2edgelist(gnnagg_handle h, int *d_edgelist);

/* Aggregator_GAT::run / run_with_feat, aggr_gat.h:317-394.  att is [V,heads,2] (heads = 1 is the
 * reference layout), x,y are [V,feat], feat % heads == 0.  d_newval (may be NULL) receives the
 * un-normalised edge weights [E,heads] the reference's scheduled kernel materialises (:187). */
int gnnagg_gat_run(gnnagg_handle h, const float *d_x, const float *d_att, float *d_y, int feat, int heads,
                   float slope, int mode, float *d_newval);
/* The fused GAT aggregation in TWO passes over disjoint edge sets of the same rows (two handles over the same rows: the
 * row-partitioned step's local-source edges, then its halo-source edges once the exchange has landed).  GNNAGG_MODE_BALANCED on
 * the chunked plan; 16-byte aligned rows of at most 256 columns.
 *   part = 1  d_y[row, :] receives the NUMERATOR sum_e w_e x_e and d_den_io[row, h] the denominator sum_e w_e; no division
 *   part = 2  both are added to what part 1 left (old + new, one fp32 add per element), then the row is divided
 *             (scaleArray, aggr_gat.h:207-213); rows this handle has no edges for are divided all the same
 *   part = 3  a pass in between (staged halo exchange, gnnagg_dist_step_create_staged): both are added, nothing is divided */
int gnnagg_gat_run_part(gnnagg_handle h, const float *d_x, const float *d_att, float *d_y, int feat, int heads, float slope, int part,
                        float *d_den_io);
/* Aggregator_GAT::run_att, aggr_gat.h:395-401 (attGat :5-31): out_val[E,heads] = softmax weights */
int gnnagg_gat_run_att(gnnagg_handle h, const float *d_att, float *d_out_val, int heads, float slope);
/* aggr_gat.h:402-425, single head as in the reference */
int gnnagg_gat_run_u_add_v(gnnagg_handle h, const float *d_att, float *d_out_val);
int gnnagg_gat_run_add_to_center(gnnagg_handle h, const float *d_in_val, float *d_out_att);
int gnnagg_gat_run_div_each(gnnagg_handle h, const float *d_in_att, float *d_inout_val);

/* spmm<L>, spmm.h:223-265 (naive thread-per-row baseline; empty rows are left untouched) */
int gnnagg_spmm_naive(const int *d_ptr, const int *d_idx, const float *d_val, const float *d_x, float *d_y,
                      int num_v, int feat, void *hip_stream);
/* valid(), spmm.h:35-69 (validate2 :11-21): number of elements with |(ref-ans)/ref| > 1e-2 */
int gnnagg_validate(const float *d_ref, const float *d_ans, int num, int *h_diff, void *hip_stream);
/* validReordered(), spmm.h:71-91 (validateReordered :23-33) */
int gnnagg_validate_reordered(const float *d_ref, const float *d_ans, const int *d_map, int num_v, int feat,
                              int *h_diff, void *hip_stream);

/* ---------------------------------------------------------------------------------------------
 * C. Host graph preparation (pure host code; usable without a GPU)
 * ------------------------------------------------------------------------------------------- */
/* load_graph, src/data.cu:31-139.  Reads <datadir><dset>.config/.graph (or the .ptrdump/.edgedump
 * caches, which it also writes), applies <datadir><dset>.reorder<reorder_suffix> when shuffle != 0
 * and the file exists.  Unlike the reference (data.cu:34 hard-codes "../data/") datadir is honoured.
 * Outputs are malloc'ed; release with gnnagg_free_host.  rows / reverse_rows are NULL outputs when
 * no reorder was applied. */
int gnnagg_load_graph(const char *datadir, const char *dset, const char *reorder_suffix, int shuffle,
                      int *num_v, int *num_e, int **h_ptr, int **h_idx, int **h_rows, int **h_reverse_rows);
void gnnagg_free_host(void *p);
/* reorderCSR, src/data.cu:4-29 */
int gnnagg_reorder_csr(const int *h_ptr, const int *h_idx, const int *h_map, const int *h_reverse_map,
                       int num_v, int num_e, int *h_newptr, int *h_newidx);
/* neighbor_grouping_schedule, graph_schedule.h:91-126.  Outputs may be NULL to size first. */
int gnnagg_neighbor_grouping_schedule(const int *h_ptr, int neighbor_num, int num_v, int *h_ptr_out,
                                      int *h_target_out, int *num_groups);
/* locality_schedule :17-63 (neighbor_num <= 0) / localityNeighborGrouping :156-211 */
int gnnagg_locality_schedule(const int *h_ptr, const int *h_idx, const float *h_val, int par_num,
                             int neighbor_num, int num_v, int total_num_v, int *h_ptr_out, int *h_idx_out,
                             float *h_val_out, int *h_target_out, int *num_groups);

/* Locality reorder GENERATOR: the clustering of script/cluster2.py:29-171 (MinHash with num_perm
 * permutations, LSH at Jaccard `threshold`, greedy union-find merge of the most similar candidate rows
 * until clusters reach cluster_cap nodes).  0 / 0.0 select the reference's constants (64, 0.2, 64).
 * h_rows_out[i] = old node id placed at new position i: the content of a <dset>.reorder<suffix> file. */
int gnnagg_cluster_reorder(const int *h_ptr, const int *h_idx, int num_v, float threshold, int num_perm, int cluster_cap,
                           unsigned long long seed, int *h_rows_out, int *num_clusters);
/* The same clustering with a choice of the order the clusters are written in.  order_mode 0: by first member (the
 * reference script, cluster2.py:156-171; = gnnagg_cluster_reorder).  order_mode 1: cache-aware greedy -- each next cluster
 * is the one whose rows find the largest share of their source rows among the cache_rows most recently gathered rows (an
 * LRU model of one XCD's L2; 0 = 4096 rows, 2 MB of 512-byte feature rows), so clusters with overlapping neighbor sets
 * become neighbors in the new numbering.  Deterministic: the rows are a function of the arguments (graphs of 20 M edges and more
 * are walked by 64 logical walkers in bulk-synchronous rounds -- GNNAGG_REORDER_WALKERS overrides the count, never the thread count). */
int gnnagg_cluster_reorder_ex(const int *h_ptr, const int *h_idx, int num_v, float threshold, int num_perm, int cluster_cap,
                              unsigned long long seed, int order_mode, int cache_rows, int *h_rows_out, int *num_clusters);

/* ---------------------------------------------------------------------------------------------
 * D. 1-D row partition + halo exchange support
 * ------------------------------------------------------------------------------------------- */
/* nnz-balanced contiguous row blocks: bounds[nparts+1], bounds[0]=0, bounds[nparts]=num_v. */
int gnnagg_partition_rows(const int *h_ptr, int num_v, int nparts, int *h_bounds);
/* For rank `rank` owning rows [bounds[rank], bounds[rank+1]): builds the local CSR whose column ids
 * are local-X slots: owned columns -> col - bounds[rank]; remote columns -> n_local + halo slot,
 * halo slots grouped by owner rank in ascending global id.  h_local_ptr[n_local+1] and
 * h_local_idx[nnz_local] are caller-allocated; h_halo_ids (malloc'ed, gnnagg_free_host) lists the
 * global ids of the halo slots; h_halo_counts[nparts] = halo ids owned by each rank. */
int gnnagg_halo_plan(const int *h_ptr, const int *h_idx, int num_v, const int *h_bounds, int nparts, int rank,
                     int *h_local_ptr, int *h_local_idx, int **h_halo_ids, int *h_halo_counts, int *num_halo);
/* The same plan from the rank's OWN rows only: h_ptr_slice[0 .. n_local] (any base offset: the slice of a global ptr array
 * or a 0-based local one) and h_idx_slice[nnz_local] with GLOBAL column ids; num_cols = global column count.  Nothing of
 * the other ranks' rows is needed (gnnagg_halo_plan reads the same data out of a global CSR). */
int gnnagg_halo_plan_slice(const int *h_ptr_slice, const int *h_idx_slice, int num_cols, const int *h_bounds, int nparts, int rank,
                           int *h_local_ptr, int *h_local_idx, int **h_halo_ids, int *h_halo_counts, int *num_halo);
/* The stage plan of a staged halo exchange (gnnagg_dist_step_create_staged): how the rows of every (reader <- owner) list are dealt
 * to stages.  GNNAGG_STAGES_STRIPE: every stage takes slice j of k of EVERY list (each stage keeps all links of the xGMI mesh busy);
 * GNNAGG_STAGES_OWNER: stage s carries the whole lists of the pairs at ring distance s + 1 (world - 1 stages, one peer per stage).
 * Both ends of a pair cut their common list by the same rule, so no coordination is needed.
 *   receiving side (h_recv_rows[world], rows per owner in gnnagg_halo_plan's owner-major slot order; NULL to skip):
 *     h_stage_recv[n_stages][world], h_new_of_old[n_halo] (may be NULL) = the stage-major slot of every owner-major slot: renumber
 *     the halo columns of the local CSR and the halo id list with it; requests still go out owner-major, every owner's list in
 *     the order its rows will ARRIVE (stage by stage)
 *   sending side (h_send_rows[world], rows per reader as served reader-major; NULL to skip):
 *     h_stage_send[n_stages][world], h_send_order[n_send] (may be NULL) = for every stage-major position of the send buffer the index
 *     into the reader-major serve list
 * *n_stages is always written (call with both lists NULL to size the outputs). */
#define GNNAGG_STAGES_STRIPE 0
#define GNNAGG_STAGES_OWNER 1
int gnnagg_halo_stage_plan(const long long *h_recv_rows, const long long *h_send_rows, int world, int rank, int mode, int k, int *n_stages,
                           long long *h_stage_recv, int *h_new_of_old, long long *h_stage_send, int *h_send_order);
/* out[i,:] = x[ids[i],:] for i < n  (send-buffer pack before the all-to-all) */
int gnnagg_pack_rows(const float *d_x, const int *d_ids, int n, int feat, float *d_out, void *hip_stream);

/* RCCL transport (one process per GPU; SURVEY.md 8e: grouped ncclSend / ncclRecv over the xGMI mesh).  librccl is loaded
 * on first use (GNNAGG_ERR_STATE when it cannot be); the communicator binds to the calling thread's current HIP device.
 *   gnnagg_dist_unique_id            rank 0 creates the 128-byte id; the launcher hands it to the other ranks (a file, an
 *                                    environment variable, MPI, a torch.distributed broadcast ...)
 *   gnnagg_dist_comm_create          every rank, same id (collective: returns when all `world` ranks have called it)
 *   gnnagg_dist_comm_create_from_file  the same with the id passed through `path`: all a C++ driver started once per GPU needs.
 *                                    Ranks > 0 publish a fresh random token (<path>.req.<rank>); rank 0 removes whatever an earlier
 *                                    launch left at `path`, draws the id and publishes {magic, world, id, tokens} atomically; a
 *                                    rank accepts only a file that carries ITS token, so a stale file from a crashed run can never
 *                                    feed mismatched ids to ncclCommInitRank (which would hang).  Every wait is bounded by
 *                                    timeout_s (120 when <= 0).  `path` should still be unique per launch (two concurrent
 *                                    launches on one path would fight over it)
 *   gnnagg_dist_alltoallv            h_send_counts[p] / h_recv_counts[p] elements of elem_bytes bytes to / from rank p,
 *                                    packed contiguously in rank order in d_send / d_recv; asynchronous on hip_stream
 *   gnnagg_dist_halo_exchange        one aggregation's halo pull: packs x_local[send_ids] (rows the peers asked for, in
 *                                    rank order) into d_send_buf and exchanges rows of feat floats into d_x_halo */
#define GNNAGG_UNIQUE_ID_BYTES 128
typedef int64_t gnnagg_comm;
int gnnagg_dist_unique_id(char *id128);
int gnnagg_dist_comm_create(const char *id128, int rank, int world, gnnagg_comm *out);
int gnnagg_dist_comm_create_from_file(const char *path, int rank, int world, int timeout_s, gnnagg_comm *out);
int gnnagg_dist_comm_destroy(gnnagg_comm c);
int gnnagg_dist_comm_info(gnnagg_comm c, int *rank, int *world);
/* What really carries this communicator's messages, so that a measurement can say it: the file the nccl* entry points were
 * loaded from (dladdr), whether that was the GNNAGG_RCCL_LIB override (a test double: ranks as processes on one GPU) and the PCI bus id
 * of the HIP device the communicator is bound to (ranks all-gather it: fewer distinct ids than ranks = ranks sharing a GPU).  Any
 * output may be NULL; strings are NUL-terminated and truncated to their capacity. */
int gnnagg_dist_transport_info(gnnagg_comm c, char *library_path, int library_path_cap, char *pci_bus_id, int pci_bus_id_cap, int *is_override);
int gnnagg_dist_alltoallv(gnnagg_comm c, const void *d_send, const long long *h_send_counts, void *d_recv,
                          const long long *h_recv_counts, int elem_bytes, void *hip_stream);
int gnnagg_dist_halo_exchange(gnnagg_comm c, const float *d_x_local, const int *d_send_ids, const long long *h_send_rows,
                              const long long *h_recv_rows, int feat, float *d_send_buf, float *d_x_halo, void *hip_stream);

/* GAT halo rows travel with their attention terms in ONE exchange: out[i, :] = [x[ids[i], 0 .. feat) | att[ids[i], 0 .. att_width)]
 * (att_width = 2 * heads), and the receiving side's split into the halo tails of X_ext / att_ext. */
int gnnagg_pack_rows2(const float *d_x, const float *d_att, const int *d_ids, int n, int feat, int att_width, float *d_out, void *hip_stream);
int gnnagg_unpack_rows2(const float *d_in, int n, int feat, int att_width, float *d_x_out, float *d_att_out, void *hip_stream);

/* The whole row-partitioned step behind ONE host call (pack -> grouped send / recv on a communication stream ->
 * local-source pass -> event wait -> halo-source pass).  Asynchronous: stream operations only (fork / join of the caller's stream
 * and the step's own communication stream through events), nothing allocated per step, so a warm step can be captured into a HIP
 * graph.  A rank without peers (comm = 0 or world 1) or without halo rows never creates the second stream.
 *   gnnagg_dist_step_create   comm (0: single rank); agg_local = aggregator over the edges whose source is an owned row, agg_remote
 *                             = aggregator over the halo-source edges (0: none); d_send_ids / h_send_rows / h_recv_rows = the plan
 *                             of gnnagg_dist_halo_exchange (the counts are copied)
 *   gnnagg_dist_step_gcn      y = A_loc . x_local, then y += A_rem . x_halo (GNNAGG_FLAG_ACCUMULATE); agg_remote's column ids are
 *                             halo slots (0-based); reduce = mean / max need gnnagg_set_row_aux on the aggregators (total degrees on
 *                             both for mean; local-source degrees on agg_remote for max)
 *   gnnagg_dist_step_gat      x_ext = [X_local ; X_halo], att_ext likewise ([., heads, 2]); ONE exchange carries [x | att] rows
 *                             (d_send_buf [n_send, feat + 2 heads], d_recv_buf [n_halo, feat + 2 heads]); both aggregators index
 *                             X_ext slots; d_den [n_local, heads] carries the denominators between the two passes
 *                             (gnnagg_gat_run_part)
 *   gnnagg_dist_step_create_staged  the exchange PIPELINED against the halo-source pass: the halo rows arrive in n_stages stages,
 *                             stage s with its own grouped send / recv (h_send_rows / h_recv_rows: [n_stages][world], rows) and its
 *                             own event, and agg_remote[s] (0: no edges) -- an aggregator over the halo-source edges whose source
 *                             arrives in stage s -- runs as soon as stage s has landed, while stage s + 1 is in flight.  Buffers are
 *                             STAGE-MAJOR: d_send_ids, d_send_buf, the halo tail and d_recv_buf list stage 0's rows in rank order,
 *                             then stage 1's ...; agg_remote[s] indexes the whole halo tail (GCN: 0-based halo slots; GAT: X_ext
 *                             slots).  How the rows are dealt to stages is the caller's plan (gnn_computing_amd/dist.py:
 *                             "stripe" -- every stage takes 1/K of EVERY peer's rows, so each stage keeps all xGMI links of the
 *                             mesh busy; "owner" -- stage s exchanges with the peers at ring distance s + 1 only).  The result
 *                             is y = local pass, then += stage 0's pass, += stage 1's ... in this fixed order: deterministic.
 *                             mean: total degrees on every aggregator; max: agg_remote[s] gets the edges folded BEFORE stage s */
typedef int64_t gnnagg_dist_step_t;
int gnnagg_dist_step_create(gnnagg_comm comm, gnnagg_handle agg_local, gnnagg_handle agg_remote, const int *d_send_ids,
                            const long long *h_send_rows, const long long *h_recv_rows, gnnagg_dist_step_t *out);
int gnnagg_dist_step_create_staged(gnnagg_comm comm, gnnagg_handle agg_local, int n_stages, const gnnagg_handle *agg_remote,
                                   const int *d_send_ids, const long long *h_send_rows, const long long *h_recv_rows,
                                   gnnagg_dist_step_t *out);
/* (n_stages, the world size of the step's communicator -- 1 without one) */
int gnnagg_dist_step_info(gnnagg_dist_step_t step, int *n_stages, int *world);
int gnnagg_dist_step_destroy(gnnagg_dist_step_t step);
int gnnagg_dist_step_gcn(gnnagg_dist_step_t step, const float *d_x_local, float *d_x_halo, float *d_send_buf, float *d_y, int feat, int reduce,
                         void *hip_stream);
int gnnagg_dist_step_gat(gnnagg_dist_step_t step, float *d_x_ext, float *d_att_ext, int n_local, float *d_send_buf, float *d_recv_buf,
                         float *d_den, float *d_y, int feat, int heads, float slope, void *hip_stream);

/* ---------------------------------------------------------------------------------------------
 * E. Extras: libgnnagg_extras.so only (`make -C gnn_computing_amd/csrc extras`, -DGNNAGG_EXTRAS).  NOT in the shipped library:
 *    the reference is forward-only and SURVEY 2.2 marks its one backward kernel ("Experiment", no caller) out of scope.
 * ------------------------------------------------------------------------------------------- */
#ifdef GNNAGG_EXTRAS
/* Backward of the GCN / SAGE sum aggregation y = A.x (no reference counterpart: the reference is forward-only):
 * d_dinput[V, feat] = A^T . d_doutput with the aggregator's edge values, as a deterministic gather over the
 * transposed CSR (built once per handle) on the balanced kernels.  Square graphs (neighbor ids < num_v). */
int gnnagg_gcn_run_bwd(gnnagg_handle h, const float *d_doutput, float *d_dinput, int feat);
/* Aggregator_GAT::run_bwd, aggr_gat.h:426-434 (kernel aggr_gat_fine_bwd :222-296, marked "Experiment" in the reference and
 * called by none of its drivers).  Backward of the single-head fused aggregation out_r = sum_e w_e x_s / D_r given the
 * forward pass's un-normalised edge weights newval[E] (w_e) and denominators div[V] (D_r):
 *   d_feat_grad[V,feat]  = gradient w.r.t. the input features through the aggregation (attention held constant)
 *   d_a_b_grad[V,2]      = gradient w.r.t. att: [.,0] centre term, [.,1] source term
 * Differences from the reference kernel, which stops short of its own comments: all `feat` columns (not 32), the
 * leaky-ReLU slope applied where z_e < 0 (the reference tests newval < 0, never true), the centre-term gradient is
 * produced, and both outputs are OVERWRITTEN with deterministic sums (the reference atomically accumulates into
 * caller-zeroed buffers).  heads == 1 only. */
int gnnagg_gat_run_bwd(gnnagg_handle h, const float *d_output, const float *d_doutput, const float *d_newval, const float *d_div,
                       const float *d_infeat, float *d_a_b_grad, float *d_feat_grad, float relu_slope, int feat);
#endif /* GNNAGG_EXTRAS */

#ifdef __cplusplus
}
#endif
#endif /* GNNAGG_H */
