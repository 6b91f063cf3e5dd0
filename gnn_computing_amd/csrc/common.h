// common.h -- internal declarations shared by the translation units of libgnnagg.so.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>

#include "../../include/gnnagg.h"

namespace gnnagg {

// records the message for gnnagg_last_error() and returns `code`
int fail(int code, const std::string &msg);

// ---- host_graph.cpp
void reorder_csr(const int *ptr, const int *idx, const int *map, const int *rmap, int V, int *newptr, int *newidx);
int neighbor_grouping(const int *ptr, int ng, int V, int *ptr_out, int *target_out);
int locality_schedule(const int *ptr, const int *idx, const float *val, int par_num, int ng, int V, int total_v,
                      int *ptr_out, int *idx_out, float *val_out, int *target_out, int *eid_out = nullptr);  // eid_out[pos] = original edge of permuted position pos
int load_graph(const char *datadir, const char *dset, const char *suffix, int shuffle, int *num_v, int *num_e,
               int **ptr_o, int **idx_o, int **rows_o, int **rrows_o);
void partition_rows(const int *ptr, int V, int nparts, int *bounds);
int halo_plan_slice(const int *ptr_slice, const int *idx_slice, int num_cols, const int *bounds, int nparts, int rank, int *lptr,
                    int *lidx, int **halo_ids_o, int *halo_counts, int *num_halo);
int halo_plan(const int *ptr, const int *idx, int V, const int *bounds, int nparts, int rank, int *lptr, int *lidx,
              int **halo_ids_o, int *halo_counts, int *num_halo);
int halo_stage_count(int world, int mode, int k);
void halo_stage_plan_recv(const long long *recv_rows, int world, int rank, int mode, int k, long long *stage_recv, int *new_of_old);
void halo_stage_plan_send(const long long *send_rows, int world, int rank, int mode, int k, long long *stage_send, int *order);

// ---- reorder.cpp
int cluster_reorder(const int *ptr, const int *idx, int V, double threshold, int num_perm, int cap, uint64_t seed,
                    int max_bucket, int *rows_out, int *num_clusters_out, int order_mode = 0, int cache_rows = 4096);

// ---- agg_gcn.hip / agg_gat.hip / aux_kernels.hip : launch descriptors (all pointers are device pointers)

// A work list: item g covers edges [ptr[g], ptr[g+1]) of output row (target ? target[g] : g).
//   slot == nullptr : every item owns its whole row -> direct store.
//   slot[g] <  0    : item owns the whole row -> direct store
//   slot[g] >= 0    : row is split over several items -> partial sums go to scratch row slot[g]
// Items [n_items, n_items + n_empty) zero-fill the rows listed in empty_rows (rows with no item).
// Split rows with more partial rows than this are combined by a whole workgroup (LDS-staged), the others by one lane group
// in batches of 16 loads.  (The source-partitioned order gives EVERY row one partial per range: 16 or 32 of them.)
static constexpr int kBigRowPartials = 64;

struct WorkList {
    const int *ptr = nullptr;
    const int *target = nullptr;
    const int *slot = nullptr;
    const int *empty_rows = nullptr;
    int n_items = 0;
    int n_empty = 0;
    // multi-item rows: row mrow_id[m] sums scratch rows [mrow_ptr[m], mrow_ptr[m+1]) in order
    const int *mrow_id = nullptr;
    const int *mrow_ptr = nullptr;
    int n_mrows = 0;
    int n_slots = 0;
    // multi-item rows with more partials than one lane group keeps in flight (hub rows): indices into mrow_*
    const int *big_rows = nullptr;
    int n_big = 0;
};

// Lane geometry of the 2-D blocked mode: 16-byte lanes on a tile of `tile_w` floats (re-tiled or line-aligned X).
struct TileSpec {
    int on = 0;            // 1: tile-major launch with the strides below
    int tile_w = 64;       // floats per column tile (GROUP = tile_w / 4 lanes)
    int xpitch = 0;        // floats between consecutive X rows as the kernel sees them
    long x_tile_stride = 0;  // floats between the first columns of consecutive tiles of X
    int ppitch = 0;        // floats between consecutive partial rows
    long p_tile_stride = 0;  // floats between consecutive tiles of the partial scratch
    int yvec = 1;          // alignment class of the caller's Y rows (4 / 2 / 1)
};

struct GcnLaunch {
    WorkList wl;
    const int *row_ptr = nullptr;  // original CSR ptr (degrees for mean)
    const int *idx = nullptr;
    const float *val = nullptr;  // nullptr => implicit 1
    const float *x = nullptr;
    float *y = nullptr;
    float *partial = nullptr;  // [n_slots, F] scratch
    int feat = 0;
    int reduce = GNNAGG_REDUCE_SUM;
    int xcd_remap = 1;
    int accumulate = 0;  // combine only: y += sum of partials
    const int *row_aux = nullptr;
    int relu = 0;        // y = max(result, 0)
    void *timer = nullptr;          // run_clock: unsigned long long[3 * blocks]
    int *timer_blocks_out = nullptr;  // run_clock: receives the number of workgroups of the items kernel
    int timer_capacity = 0;           // run_clock: workgroups the timer buffer has room for (checked against the grid)
    // host array [n_items + n_empty + 1]: prefix sums of the per-item cost, for xcd_remap == 2
    const long *xcd_item_cost_prefix = nullptr;
};

// Balanced plan (GNNAGG_MODE_BALANCED, GCN): see k_gcn_plan in agg_gcn.hip.
struct GcnPlanLaunch {
    const void *t0 = nullptr;  // int4 {beg,end,row,-} per short row
    const void *t1 = nullptr;  // int4 {beg,end,dest,-} per long-row segment
    int n0 = 0, n1 = 0, chunk = 64;
    const long *t0_cost_prefix = nullptr;  // host, n0+1 entries
    WorkList hubs;                          // only mrow_* / big_rows / n_slots are used (combine of multi-segment rows)
    const int *row_ptr = nullptr;
    const int *idx = nullptr;
    const float *val = nullptr;
    const float *x = nullptr;
    float *y = nullptr;
    float *partial = nullptr;
    int feat = 0;
    int reduce = GNNAGG_REDUCE_SUM;
    int xcd_remap = 2;
    int accumulate = 0;  // y += A.x (sum; mean / max with row_aux); rows without edges keep their value
    const int *row_aux = nullptr;  // gnnagg_set_row_aux: mean divisor / edges already folded into y (finish_gcn_row)
    int relu = 0;        // y = max(result, 0)
    int num_rows = 0;    // rows of y
    int t0_partials = 0; // short-row descriptors may carry scratch slots (dest < 0): source-partitioned order
    // hubs finished inside the plan kernel by the last segment workgroup to arrive (no k_combine launch)
    const int *slot_hub = nullptr;  // device: scratch slot -> index into hubs.mrow_*
    int *hub_count = nullptr;       // device: arrival counters, zero between launches; n_mrows * hub_count_stride ints
    int hub_count_stride = 0;       // counters per hub (>= column tiles of the launch)
    // dense combine as the epilogue (run_with_nn): nn_out[V, nn_cols] = y . nn_weight[feat, nn_cols]
    const float *nn_weight = nullptr;
    float *nn_out = nullptr;
    int nn_cols = 0;
    TileSpec tile;  // 2-D blocked mode (short-row descriptors only: n1 == 0)
    int probe = 0;  // 1: gather probe -- the same descriptors, id/value loads and feature gathers, no chain, no stores
    int unroll = 0; // 4: four gathers per batch where the geometry has that instantiation (balanced / scheduled orders); 0: default (8)
};

// 2-D blocked order as a segmented stream (agg_span.hip): lane groups walk spans of whole groups of the permuted edge list.
struct SpanLaunch {
    const int *span_g = nullptr;              // [n_spans + 1] first group of every span
    int n_spans = 0;
    const long *span_cost_prefix = nullptr;   // host, n_spans + 1 entries: edges before every span
    const int *ptr_s = nullptr;               // [n_groups + 1] edge offsets of the groups (permuted order)
    const int *idx_f = nullptr;               // ids with the group-end flags in the top two bits
    const float *val_s = nullptr;             // permuted values, nullptr => implicit 1
    const int *target = nullptr;              // [n_groups] row of every group
    int n_groups = 0;
    const int *crows = nullptr;               // rows with >= 2 groups, most groups first
    int n_crows = 0;
    const int *rg_ptr = nullptr, *rg_idx = nullptr;  // row -> its groups (ascending)
    const int *empty_rows = nullptr;          // rows without any group
    int n_empty = 0;
    const int *row_ptr = nullptr;             // CSR ptr (degrees for mean)
    const float *x = nullptr;                 // the image of X named by `tile`
    int x_rows = 0;                           // rows of that image (ids are < x_rows)
    float *y = nullptr;
    float *partial = nullptr;                 // [ntiles][n_groups][tile_w]
    int feat = 0, reduce = GNNAGG_REDUCE_SUM, relu = 0;
    TileSpec tile;
    int probe = 0;
    // chain = 1 (canonical rows mode on the blocked order): the spans of ONE source range; every group's chain starts from and
    // returns to partial = Yt[ntiles][n_groups = V rows][tile_w]; no combine, no zero-fill (launch_untile_y finishes)
    int chain = 0;
};
int launch_gcn_span(const SpanLaunch &a, void *stream);
// y[r, :] = finish(Yt[tile][r][:]) -- the tiled image back into the caller's rows, mean (/ degree) and ReLU applied
int launch_zero_words(void *p, size_t n_words, void *stream);   // 4-byte words; a kernel, not a memset (graph replays: aux_kernels.hip)
// (skip: optional [rows] bytes, 1 = leave the row of y alone)
int launch_untile_y(const float *yt, float *y, const int *row_ptr, const unsigned char *skip, int rows, int feat, int tile_w, int mean, int relu,
                    void *stream);
// GAT flavour: y[r, c] = Yt[tile][r][c] / den_t[tile][r][head of c inside the tile] (0 where the denominator is 0: rows without edges)
int launch_untile_y_gat(const float *yt, const float *den_t, float *y, const unsigned char *skip, int rows, int feat, int tile_w, int ht, int dhead,
                        void *stream);
struct GatSpanLaunch {
    SpanLaunch s;                   // val_s unused
    const float *att = nullptr;     // [V, H, 2]
    const float *as_t = nullptr;    // compact source terms  [ceil(H / HT)][att_rows][HT] (k_tile_att), HT = heads per tile
    const float *ac_t = nullptr;    // compact centre terms, same layout
    int att_rows = 0;
    float *partial_den = nullptr;   // [n_groups, H]
    float *newval = nullptr;        // optional [E, H], CSR edge order (scattered through eperm)
    const int *eperm = nullptr;
    int heads = 1;
    float slope = 0.2f;
    float *den_t = nullptr;         // s.chain = 1: per-tile denominator image [ntiles][den_rows][HT] (the numerator image is s.partial)
    int den_rows = 0;
};
int launch_gat_span(const GatSpanLaunch &a, void *stream);
// whether the GAT span kernel covers (feat, heads) at this tile width
static inline bool gat_span_tiles(int feat, int heads, int tile_w)
{
    if (heads <= 0 || feat % heads != 0) return false;
    const int dhead = feat / heads;
    if (dhead % 4 != 0) return false;
    if (tile_w >= dhead) { const int ht = tile_w / dhead; return tile_w % dhead == 0 && (ht == 1 || ht == 2 || ht == 4 || ht == 8); }
    return dhead % tile_w == 0;
}

// Long rows of the rows mode (`scheduled = 0`, canonical CSR-order chains): k_gcn_rows_long.
struct GcnRowsLongLaunch {
    const void *r1 = nullptr;  // int4 {beg,end,row,-} per long row, heaviest first
    int n1 = 0;
    const int *idx = nullptr;
    const float *val = nullptr;
    const float *x = nullptr;
    float *y = nullptr;
    int feat = 0;
    int reduce = GNNAGG_REDUCE_SUM;
    int relu = 0;  // y = max(result, 0) (GCN flavour)
    const float *att = nullptr;  // non-null: GAT flavour (fused edge softmax), needs (feat / heads) % 32 == 0
    int heads = 1;
    float slope = 0.2f;
    int medium = 0;  // 1: the 128-thread form (rows of the medium class: many workgroups per CU)
    int tile_w = 0;  // hub form, GCN flavours: 0 = the launcher's rule, 32 / 64 = that column-tile width ("rows_hub_tile")
};

struct GatLaunch {
    WorkList wl;
    const int *idx = nullptr;
    const float *att = nullptr;  // [V,H,2]
    const float *x = nullptr;
    float *y = nullptr;
    float *partial = nullptr;      // [n_slots, F]
    float *partial_den = nullptr;  // [n_slots, H]
    float *newval = nullptr;       // optional [E,H] un-normalised weights
    int feat = 0;
    int heads = 1;
    float slope = 0.2f;
    int xcd_remap = 1;
};

// Edge kernels on chunked work items (attGat / u_add_v / add_to_center / each_div without the hub-row tail)
struct EdgeItemLaunch {
    WorkList wl;
    const int *idx = nullptr;
    const float *att = nullptr;
    const float *in = nullptr;
    float *out = nullptr;
    float *den = nullptr;
    float *partial_den = nullptr;
    int heads = 1;
    float slope = 0.2f;
    int avg_item_edges = 8;
};
int launch_edge_items_sum(const EdgeItemLaunch &L, int op, void *stream);
int launch_edge_items_map(const EdgeItemLaunch &L, int op, void *stream);
// Balanced plan, GAT (k_gat_plan).  t1 descriptors carry the destination row in .w (the segment's attention centre).
struct GatPlanLaunch {
    const void *t0 = nullptr, *t1 = nullptr;
    int n0 = 0, n1 = 0, chunk = 64;
    const long *t0_cost_prefix = nullptr;
    WorkList hubs;
    const int *idx = nullptr;
    const float *att = nullptr;
    const float *x = nullptr;
    float *y = nullptr;
    float *partial = nullptr, *partial_den = nullptr, *newval = nullptr;
    int feat = 0, heads = 1;
    float slope = 0.2f;
    int xcd_remap = 2;
    int rows_semantics = 0;  // 1: `scheduled = 0` semantics (aggr_gat: divide by the denominator unconditionally)
    // hubs folded inside the plan kernel (see GcnPlanLaunch)
    const int *slot_hub = nullptr;
    int *hub_count = nullptr;
    int hub_count_stride = 0;
    TileSpec tile;              // 2-D blocked mode, as in GcnPlanLaunch
    const int *eperm = nullptr; // permuted orders: original edge of every position (newval is written in CSR edge order)
    int unroll = 0;             // 4: four gathers per batch where the geometry has that instantiation (balanced / scheduled orders)
    int part_mode = 0;          // two-pass form (gnnagg_gat_run_part): 1 = numerator / denominator out, 2 = add to them and divide
    float *den_io = nullptr;    // [num_v, heads]
};
int launch_gat_plan(const GatPlanLaunch &a, void *stream);
// Backward of the single-head fused GAT aggregation (k_rowdot + k_gat_bwd_edges); wl = chunked edge work items.
struct GatBwdLaunch {
    WorkList wl;
    const int *idx = nullptr;
    const float *out = nullptr, *dout = nullptr, *newval = nullptr, *div = nullptr, *x = nullptr;
    float *rowdot = nullptr, *dz = nullptr;
    int V = 0, feat = 0;
    float slope = 0.2f;
};
int launch_gat_bwd_edges(const GatBwdLaunch &a, void *stream);
int launch_gat_bwd_permute(const int *perm, const int *idx_t, const float *dz, const float *newval, const float *div, float *dz_t,
                           float *val_t, int E, void *stream);
int launch_interleave2(const float *a, const float *b, float *out, int n, void *stream);
int launch_permute_val(const int *perm, const float *val, float *val_t, int E, void *stream);
int launch_gcn(const GcnLaunch &a, void *stream);
int launch_gcn_plan(const GcnPlanLaunch &a, void *stream);
int launch_gcn_rows_long(const GcnRowsLongLaunch &a, void *stream);
int launch_gat(const GatLaunch &a, void *stream);
int launch_csr2edgelist(const int *ptr, const int *idx, int *edgelist, int V, int avg_deg, void *stream);
int launch_edgewise(const int *edgelist, const float *val, const float *x, float *y, int E, int V, int feat,
                    void *stream);
int launch_spmm_naive(const int *ptr, const int *idx, const float *val, const float *x, float *y, int V, int feat,
                      void *stream);
int launch_validate(const float *ref, const float *ans, int num, int *d_diff, void *stream);
int launch_validate_reordered(const float *ref, const float *ans, const int *map, int V, int feat, int *d_diff,
                              void *stream);
int launch_dense_rows(const int *rows, int n_rows, const float *Y, const float *W, float *out, int K, int N, void *stream);
int launch_dense_nn(const float *A, const float *B, float *C, int M, int N, int K, void *stream);
int launch_check_csr(const int *ptr, const int *idx, int V, int E, int num_cols, int *d_counts, void *stream);
int launch_pack_rows(const float *x, const int *ids, int n, int feat, float *out, void *stream);
int launch_pack_rows2(const float *x, const float *att, const int *ids, int n, int feat, int att_w, float *out, void *stream);
int launch_unpack_rows2(const float *in, int n, int feat, int att_w, float *x_out, float *att_out, void *stream);
// xt[t][r][0..tile_w) = x[r][t*tile_w ..] (zero beyond feat): the column-tiled image of X the 2-D blocked mode gathers from
int launch_tile_x(const float *x, float *xt, int rows, int feat, int tile_w, void *stream);
int launch_probe_row_gather(const void *rows, long pitch, int seg_bytes, const int *ids, long n_ids, int per_group, void *stream);
int launch_tile_att(const float *att, float *as_t, float *ac_t, int rows, int heads, int ht, void *stream);

}  // namespace gnnagg
