// api_internal.h -- what the translation units behind the C-ABI share: the handle object (Ctx), its schedules and plans, the handle
// registry, and the two run dispatchers.  Not installed; include/gnnagg.h is the interface.
//
// GNNAGG_EXTRAS (make -C gnn_computing_amd/csrc extras -> libgnnagg_extras.so): what the DEFAULT library leaves out --
// the backward entry points (gnnagg_gcn_run_bwd / gnnagg_gat_run_bwd: SURVEY 2.2 marks the reference's kernel out of scope), the older
// forms of the blocked order and of the hub fold ("retile", "tiled", "spans", "inkernel_combine", "host_plan", GNNAGG_PLAN=0) that exist
// only so that second-tier parity tests can compare them bit for bit with the shipped form, and the "partition_min_degree" knob.  In
// the default build the switches of those forms are compile-time constants, so their branches are not even compiled.
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "common.h"
#include "devbuf.h"

namespace gnnagg {

// One schedule = the reference's (d_ptr_scheduled, d_idx_scheduled, d_target_scheduled,
// d_val_scheduled, num_target) of aggregator.h:130-133 plus what the deterministic combine needs.
struct Schedule {
    bool valid = false;
    int kind = GNNAGG_SCHED_NOP;
    int num_target = 0;
    bool permuted = false;  // locality schedules permute idx/val; neighbor grouping aliases them
    bool gpu_built = false; // library-built blocked order made on the device (plan_gpu.hip): per-edge arrays exist on the device only
                            // (idx_f, eperm), the descriptor form (slot, mrow_*, idx_s) does not exist at all
    int n_edges_perm = 0;   // ... and its edge count
    int total_cols = 0;     // locality schedules: the column count the ranges were cut from
    int par_num = 0;        // locality schedules: the number of column ranges
    std::vector<int> h_ptr_s, h_target, h_idx_s, h_slot, h_empty;
    std::vector<int> h_eperm;  // library-built permuted schedules: host copy of eperm while a plan is being cut from it (build_rows_blocked)
    std::vector<float> h_val_s;
    DevBuf<int> ptr_s, target, slot, empty_rows, mrow_id, mrow_ptr, idx_s, big_rows;
    DevBuf<int> eperm;      // library-built permuted schedules: original edge of every permuted position (val follows its edges)
    // segmented-stream form of a library-built partitioned order (agg_span.hip)
    DevBuf<int> idx_f, span_g, crows, rg_ptr, rg_idx;
    int n_spans = 0, n_crows = 0;
    std::vector<long> span_cost_prefix;
    int n_big = 0;
    DevBuf<float> val_s;
    int n_empty = 0, n_mrows = 0, n_slots = 0;
    std::vector<long> cost_prefix;  // per work item (groups then empty-row items), for the XCD ranges

    void reset()
    {
        valid = false;
        ptr_s.release(); target.release(); slot.release(); empty_rows.release();
        mrow_id.release(); mrow_ptr.release(); idx_s.release(); val_s.release(); big_rows.release(); eperm.release(); n_big = 0;
        idx_f.release(); span_g.release(); crows.release(); rg_ptr.release(); rg_idx.release(); n_spans = n_crows = 0;
        span_cost_prefix.clear();
        h_ptr_s.clear(); h_target.clear(); h_idx_s.clear(); h_val_s.clear(); h_slot.clear(); h_empty.clear(); h_eperm.clear();
        cost_prefix.clear();
        num_target = n_empty = n_mrows = n_slots = 0;
        permuted = gpu_built = false;
        n_edges_perm = 0;
    }
    WorkList worklist() const
    {
        WorkList w;
        w.ptr = ptr_s.p; w.target = target.p; w.slot = slot.p; w.empty_rows = empty_rows.p;
        w.n_items = num_target; w.n_empty = n_empty;
        w.mrow_id = mrow_id.p; w.mrow_ptr = mrow_ptr.p; w.n_mrows = n_mrows; w.n_slots = n_slots;
        w.big_rows = big_rows.p; w.n_big = n_big;
        return w;
    }
};

// GNNAGG_MODE_BALANCED plan of a GCN aggregator (k_gcn_plan): short rows, long-row segments, hub slots.
struct BalancedPlan {
    bool valid = false;
    int chunk = 64;
    int n0 = 0, n1 = 0, n_mrows = 0, n_slots = 0, n_big = 0;
    DevBuf<int> t0, t1, mrow_id, mrow_ptr, big_rows, slot_hub;
    std::vector<long> t0_cost_prefix;
    // the same short-row descriptors, degree-sorted inside windows (built on first use by a narrow-feature run)
    std::vector<int> h_t0;
    DevBuf<int> t0_sorted;
    std::vector<long> t0s_cost_prefix;
    void reset()
    {
        valid = false;
        t0.release(); t1.release(); mrow_id.release(); mrow_ptr.release(); big_rows.release(); slot_hub.release();
        t0_cost_prefix.clear(); h_t0.clear(); t0_sorted.release(); t0s_cost_prefix.clear();
        n0 = n1 = n_mrows = n_slots = n_big = 0;
    }
};
static constexpr int kSegChunksHost = 16;  // kSegChunks in kernel_util.cuh

// GNNAGG_MODE_ROWS plan of a GCN aggregator: short rows per lane group (r0), hub rows per 512-thread workgroup (r1), the rows
// between the two per 128-thread workgroup (r2: "medium").  r1_rows lists the rows of r1 then r2 (products of the fused GEMM).
struct RowsPlan {
    bool valid = false;
    bool medium_ok = true;   // false: built without the medium class (a GAT head width the long-row kernel cannot serve)
    int n0 = 0, n1 = 0, n2 = 0, long_deg = 256, med_deg = 256;
    DevBuf<int> r0, r1, r2, r1_rows;
    std::vector<long> r0_cost_prefix;
};

static constexpr int kItemCost = 2;  // fixed per-item overhead in edge-equivalents (XCD range balancing)

struct Ctx {
    enum Kind { GCN, GAT } kind;
    int V = 0, E = 0;
    const int *d_ptr = nullptr;
    const int *d_idx = nullptr;
    const float *d_val = nullptr;
    const int *row_aux = nullptr;  // gnnagg_set_row_aux (row-partitioned mean / max: see finish_gcn_row)
    hipStream_t stream = nullptr;
    std::vector<int> h_ptr;  // host mirror, fetched on first schedule (reference ctor: aggregator.h:50)
    Schedule sched[2];       // [0] user schedule (MODE_SCHEDULED), [1] balanced (MODE_BALANCED; GAT, and the
                             //     host arrays that describe the GCN plan's summation order)
    BalancedPlan plan;       // balanced mode
    BalancedPlan plan_sched; // `scheduled = 1` with a neighbor-grouping schedule, when the plan kernel suits that NG
    BalancedPlan plan_part;  // source-partitioned balanced mode: one short-row descriptor per group of sched[1]
    // canonical rows mode on the blocked order (option "rows_blocked"; build_rows_blocked / run_rows_blocked)
    struct RowsBlocked {
        bool tried = false, ok = false;
        Schedule sched;                          // the reference's locality_schedule arrays: one group per (row, range)
        DevBuf<int> span_g, idx_f;
        DevBuf<int> r1;                          // rows with a sub-row too long for one lane group: {beg, end, row, 0} for k_gcn_rows_long
        DevBuf<unsigned char> hub_mask;          // [V] 1 for those rows (the un-tiling pass leaves their rows of y alone)
        int n1 = 0;
        std::vector<int> span0;                  // [ranges + 1] first span of every range
        std::vector<std::vector<long>> cost;     // per range: edges before every span of the range
        void reset() { tried = ok = false; sched.reset(); span_g.release(); idx_f.release(); r1.release(); hub_mask.release(); n1 = 0; span0.clear(); cost.clear(); }
    } rb;
    bool keep_h_eperm = false;   // build_locality keeps the host copy of eperm (the plan being built filters its groups)
    DevBuf<float> yt;        // tiled image of Y the chains of that mode pass through
    DevBuf<float> den_t;     // ... and, GAT, the per-tile image of the softmax denominators
    int opt_rows_blocked = 1;
    int opt_hub_tile = 0;       // "rows_hub_tile": column-tile width of the 512-thread long-row form, GCN flavours (0: the launcher's rule; 32; 64)
    int opt_rows_medium = 0;    // "rows_medium_edges": rows above this many edges (up to the hub threshold) take the 128-thread workgroups (0: library rule, -1: none)
    int opt_rb_hub_edges = 0;   // "rows_hub_edges": rows with a (row, range) sub-row above this many edges leave the chained launches (0: library rule)
    RowsPlan rows_plan;      // rows mode (GCN; GAT head widths a 32-column tile fits in)
    RowsPlan rows_plan_nomed; // ... built WITHOUT the medium class (GAT head widths the long-row kernel cannot serve): a handle whose
                             // calls alternate between the two keeps both instead of rebuilding on every call
    hipStream_t aux_stream = nullptr;  // long rows of the rows mode run here, overlapping the short rows
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    Schedule sched_edges;    // chunked work items of the edge kernels (run_att, u_add_v, add_to_center, div_each)
    DevBuf<float> den;       // [V,heads] row sums of run_att
    DevBuf<float> partial, partial_den;
    DevBuf<float> xt;      // 2-D blocked mode: column-tiled image of X, rebuilt by every run (k_tile_x)
    DevBuf<float> att_t;   // 2-D blocked GAT: compact source / centre attention terms per head group, rebuilt by every run (k_tile_att)
#ifdef GNNAGG_EXTRAS   // older forms kept for A/B parity tests (second tier): run-time switches there, constants in the default build
    int tiled = 1;         // source-partitioned balanced mode runs tile-major on the tiled image ("tiled" = 0: r01 order)
    int opt_retile = 1;        // 0: gather from the caller's X when its rows are 128-byte aligned
    int use_spans = 1;         // GCN, tiled: the segmented-stream kernel (agg_span.hip); 0: one descriptor per lane group
    int inkernel_combine = 1;  // 0: hubs through k_combine (A/B)
    int use_plan = 1;          // GCN balanced mode runs k_gcn_plan (0: items + combine, the round-1 first design)
    int part_descriptors = 1;  // run the partitioned order on the plan kernels' descriptor path (0: item kernels)
    int opt_part_min_deg = 96; // measured crossover of the blocked order against the chunked plan (profiles/r02/partition_threshold.txt)
#else
    static constexpr int tiled = 1, opt_retile = 1, use_spans = 1, inkernel_combine = 1, use_plan = 1, part_descriptors = 1, opt_part_min_deg = 96;
#endif
    // gnnagg_set_option knobs (defaults from the environment, see create())
    int opt_partitions = -1;   // -1: library decides (avg degree >= opt_part_min_deg), 0: never partition, N: N source ranges
    int opt_tile_w = 64;       // floats per column tile of the 2-D blocked mode
    int opt_slice_kb = 4096;   // target size of the X slice one XCD's L2 holds (measured optimum 4-6 MB on the reddit-shaped F=602 case)
    int opt_scratch_limit_mb = 0;  // > 0: the blocked order may not take more scratch than this (else: half of the free memory)
    // `scheduled = 0` (GNNAGG_MODE_ROWS): 0 = canonical CSR-order chains, bit-exact against a sequential loop -- the default of
    // the status-returning API; 1 = the balanced order (within 1e-5 of it) -- the default of the reference-facing surfaces
    // (flat *_impl API, class shim, pybind names: gnnagg_set_option "reference_defaults").  GNNAGG_FAST_ROWS overrides both.
    int fast_rows = 0;
    bool fast_rows_from_env = false;
    // `scheduled = 1`: 1 (default) = the balanced order -- the reference's scheduled kernels add their group partials with
    // atomicAdd (aggr_gcn.h:112, aggr_gat.h:196-203), so ANY association is one of its legal results; num_target /
    // get_schedule / mode_params(SCHEDULED) keep describing the user's groups, the order that runs is the one
    // get_schedule(BALANCED) / balanced_params describe.  0 = the user's groups folded in the restated order (what the
    // bit-exact parity tests of the scheduled mode pin).  GNNAGG_FAST_SCHEDULED / option "fast_scheduled".
    int fast_scheduled = 1;
    int use_aux_stream = 1;    // rows mode: hub rows on a second stream beside the short rows (0: same stream, one after the other)
    DevBuf<int> edgelist;  // runEdgeWise cache (aggr_gcn.h:452-453)
    int xcd_remap = 2;         // 0 identity, 1 equal-count XCD ranges, 2 work-balanced XCD ranges
    DevBuf<int> hub_count;  // arrival counters of the in-kernel hub fold (zero between launches)
    int sort_window = 2048;    // narrow features: short-row descriptors degree-sorted inside windows of this many rows
    int partitions = 0;        // > 0: the balanced mode is SOURCE-PARTITIONED (high-degree graphs, see auto_partitions)
    int no_auto_partition = 0; // set when a run found the partial-row scratch too large: the handle stays on the chunked plan
    int force_host_plan = 0;   // set when a run needed the descriptor form of the blocked order (GAT head widths the span kernel does not tile,
                               // "spans" / "tiled" = 0): the order is then built by the host builder, which makes both forms
    double rb_plan_seconds = 0.0;   // ... of the chain plan of the rows mode (build_rows_blocked)
    double plan_seconds = 0.0; // wall time of the last library-chosen plan construction (gnnagg_plan_info)
    size_t plan_bytes = 0;     // device bytes the plan's arrays hold
    std::vector<long> row_cost_prefix;  // MODE_ROWS work items
#ifdef GNNAGG_EXTRAS
    // GAT backward (run_bwd): the transposed graph -- row s of A^T lists the destination rows of the edges whose source is
    // s, in ascending original edge order; perm[e'] = original edge id -- and a GCN aggregator over it
    struct Transposed {
        bool valid = false;
        DevBuf<int> ptr_t, idx_t, perm;
        DevBuf<float> val_t, dz, dz_t, rowdot, da, db;
        gnnagg_handle agg = 0;
    } tr;
#endif
    int avg_deg() const { return V > 0 ? (int)((long)E / V) : 0; }
};

extern std::mutex g_mu;
extern std::set<Ctx *> g_live;
Ctx *lookup(gnnagg_handle h);
void die_if_abort(int rc, const char *where);   // the flat reference API aborts like the reference (util.h:82-104)

struct NnRequest {  // run_with_nn: transformed[V, cols] = y . weight[feat, cols]
    const float *weight;
    float *out;
    int cols;
};
int gcn_run(Ctx *c, const float *x, float *y, int feat, int mode, int reduce, int flags = 0, const NnRequest *nn = nullptr, int probe = 0);
int edge_launch(Ctx *c, EdgeItemLaunch &L, int heads);
int do_schedule(Ctx *c, int kind, const int *param, int total_v);
int fetch_host_ptr(Ctx *c);
// Device -> host copy of a CALLER's array, ordered behind the work on the handle's stream and complete on return.  (A blocking hipMemcpy
// runs on the null stream, which a caller's non-blocking stream is not ordered against: a CSR that stream was still writing came back
// half-written -- round 6, the GPU suite run with a non-null stream current, GNNAGG_TEST_STREAM=side.)
int copy_to_host(Ctx *c, void *dst, const void *src, size_t bytes);

}  // namespace gnnagg

#define GET_CTX(h)                                                      \
    Ctx *c = lookup(h);                                                 \
    if (!c) return fail(GNNAGG_ERR_ARG, "invalid or destroyed handle")
