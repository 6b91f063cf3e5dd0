// api.hip -- the C-ABI of libgnnagg.so (include/gnnagg.h): handle objects, schedules, dispatch.
//
// Mirrors the reference's Aggregator / Aggregator_GCN / Aggregator_GAT life cycle
// (include/aggregator.h:25-151, aggr_gcn.h:362-550, aggr_gat.h:299-441) behind plain C entry points.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <mutex>
#include <queue>
#include <set>
#include <vector>

#include "api_internal.h"

namespace gnnagg {

static thread_local std::string g_last_error;
static bool g_abort_on_error = true;

int fail(int code, const std::string &msg)
{
    g_last_error = msg;
    return code;
}

std::mutex g_mu;
std::set<Ctx *> g_live;

Ctx *lookup(gnnagg_handle h)
{
    std::lock_guard<std::mutex> lk(g_mu);
    Ctx *c = reinterpret_cast<Ctx *>(h);
    return g_live.count(c) ? c : nullptr;
}

int copy_to_host(Ctx *c, void *dst, const void *src, size_t bytes)
{
    if (bytes == 0) return GNNAGG_OK;
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return GNNAGG_OK;
}

int fetch_host_ptr(Ctx *c)
{
    if (!c->h_ptr.empty()) return GNNAGG_OK;
    c->h_ptr.resize((size_t)c->V + 1);
    if (int rc_ = copy_to_host(c, c->h_ptr.data(), c->d_ptr, ((size_t)c->V + 1) * sizeof(int))) { c->h_ptr.clear(); return rc_; }
    if (c->h_ptr[0] != 0 || c->h_ptr[c->V] != c->E) {
        c->h_ptr.clear();
        return fail(GNNAGG_ERR_ARG, "CSR ptr[0] != 0 or ptr[num_v] != num_e");
    }
    return GNNAGG_OK;
}

// From (ptr_s, target): which groups own a whole row, which rows are split, which rows have no
// group at all; then upload everything.
static int finalize_schedule(Ctx *c, Schedule &s)
{
    const int V = c->V, G = (int)s.h_target.size();
    s.num_target = G;
    std::vector<int> groups_of(V, 0);
    for (int g = 0; g < G; ++g) groups_of[s.h_target[g]]++;
    std::vector<int> base(V, -1), mrow_id, mrow_ptr(1, 0), empty;
    int nslots = 0;
    for (int r = 0; r < V; ++r) {
        if (groups_of[r] == 0) empty.push_back(r);
        if (groups_of[r] > 1) {
            base[r] = nslots;
            nslots += groups_of[r];
            mrow_id.push_back(r);
            mrow_ptr.push_back(nslots);
        }
    }
    std::vector<int> slot(G), cursor(V, 0);
    for (int g = 0; g < G; ++g) {
        const int r = s.h_target[g];
        slot[g] = base[r] < 0 ? -1 : base[r] + cursor[r]++;
    }
    s.n_empty = (int)empty.size();
    s.n_mrows = (int)mrow_id.size();
    s.n_slots = nslots;
    s.cost_prefix.assign((size_t)G + empty.size() + 1, 0);
    for (int g = 0; g < G; ++g) s.cost_prefix[g + 1] = s.cost_prefix[g] + (s.h_ptr_s[g + 1] - s.h_ptr_s[g]) + kItemCost;
    for (size_t k = 0; k < empty.size(); ++k) s.cost_prefix[G + k + 1] = s.cost_prefix[G + k] + 1;
    int rc;
    for (int g = 0; g < G; ++g)
        if (s.h_ptr_s[g + 1] <= s.h_ptr_s[g]) return fail(GNNAGG_ERR_STATE, "internal: empty work item in a schedule");
    if ((rc = s.ptr_s.upload(s.h_ptr_s))) return rc;
    if ((rc = s.target.upload(s.h_target))) return rc;
    if ((rc = s.slot.upload(slot))) return rc;
    if ((rc = s.empty_rows.upload(empty))) return rc;
    s.h_slot = slot;
    s.h_empty = empty;
    if ((rc = s.mrow_id.upload(mrow_id))) return rc;
    if ((rc = s.mrow_ptr.upload(mrow_ptr))) return rc;
    {
        std::vector<int> big;
        for (int m = 0; m < (int)mrow_id.size(); ++m)
            if (mrow_ptr[m + 1] - mrow_ptr[m] > kBigRowPartials) big.push_back(m);
        s.n_big = (int)big.size();
        if ((rc = s.big_rows.upload(big))) return rc;
    }
    if (s.permuted) {
        if ((rc = s.idx_s.upload(s.h_idx_s))) return rc;
        if (!s.h_val_s.empty() && (rc = s.val_s.upload(s.h_val_s))) return rc;
    }
    s.valid = true;
    return GNNAGG_OK;
}

static int build_grouping(Ctx *c, Schedule &s, int ng, int kind)
{
    if (ng <= 0) return fail(GNNAGG_ERR_ARG, "neighbor group size must be >= 1");
    int rc = fetch_host_ptr(c);
    if (rc) return rc;
    s.reset();
    s.kind = kind;
    const int G = neighbor_grouping(c->h_ptr.data(), ng, c->V, nullptr, nullptr);
    s.h_ptr_s.resize((size_t)G + 1);
    s.h_target.resize((size_t)G);
    neighbor_grouping(c->h_ptr.data(), ng, c->V, s.h_ptr_s.data(), s.h_target.data());
    return finalize_schedule(c, s);
}

static int parts_for_cols(const Ctx *c, long cols);
static int pick_chunk(const Ctx *c);
static double wall_seconds()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

// par_num == -1 (library-chosen order only): the range count follows from the column count (parts_for_cols)
static int build_locality(Ctx *c, Schedule &s, int par_num, int ng, int total_v, int kind, bool keep_eid)
{
    if (par_num <= 0 && !(par_num == -1 && total_v < 0)) return fail(GNNAGG_ERR_ARG, "locality partition count must be >= 1");
    int rc = fetch_host_ptr(c);
    if (rc) return rc;
    std::vector<int> h_idx((size_t)c->E);
    std::vector<float> h_val;
    if (int rc_ = copy_to_host(c, h_idx.data(), c->d_idx, (size_t)c->E * sizeof(int))) return rc_;
    if (c->d_val && c->E > 0) {
        h_val.resize((size_t)c->E);
        if (int rc_ = copy_to_host(c, h_val.data(), c->d_val, (size_t)c->E * sizeof(float))) return rc_;
    }
    if (total_v < 0) {  // library-chosen partitioning: the ranges must cover every column that occurs (the CSR need not be
                        // square: a rank's local graph indexes [X_local ; X_halo])
        int mx = 0;
        for (int v : h_idx) mx = std::max(mx, v);
        if (par_num == -1) par_num = parts_for_cols(c, (long)mx + 1);
        // a forced range count ("partitions" / GNNAGG_PARTITIONS) above the column count would make the tiling kernels read
        // total_cols rows of the caller's X and att past their ends: no more ranges than columns
        par_num = std::max(1, std::min(par_num, mx + 1));
        total_v = mx + 1;
    }
    s.par_num = par_num;
    s.reset();
    s.kind = kind;
    s.permuted = true;
    s.total_cols = total_v;
    std::vector<int> ptr_s((size_t)c->E + 2), tgt((size_t)c->E + 1);
    s.h_idx_s.resize((size_t)c->E);
    if (!h_val.empty()) s.h_val_s.resize((size_t)c->E);
    std::vector<int> eid;
    if (keep_eid) eid.resize((size_t)c->E);
    const int G = locality_schedule(c->h_ptr.data(), h_idx.data(), h_val.empty() ? nullptr : h_val.data(), par_num, ng,
                                    c->V, total_v, ptr_s.data(), s.h_idx_s.data(),
                                    h_val.empty() ? nullptr : s.h_val_s.data(), tgt.data(), keep_eid ? eid.data() : nullptr);
    ptr_s.resize((size_t)G + 1);
    tgt.resize((size_t)G);
    const int kept = ptr_s[G];
    if (keep_eid) {
        eid.resize((size_t)kept);
        int rc2 = s.eperm.upload(eid);
        if (rc2) return rc2;
        if (c->keep_h_eperm) s.h_eperm = eid;
    }
    s.h_idx_s.resize((size_t)kept);
    if (!s.h_val_s.empty()) s.h_val_s.resize((size_t)kept);
    s.h_ptr_s.swap(ptr_s);
    s.h_target.swap(tgt);
    return finalize_schedule(c, s);
}

static int build_plan_into(Ctx *c, BalancedPlan &p, int chunk, bool describe_in_sched1, double *padding_ratio)
{
    int rc = fetch_host_ptr(c);
    if (rc) return rc;
    p.reset();
    p.chunk = chunk;
    const int V = c->V;
    const long seg_edges = (long)chunk * kSegChunksHost;
    std::vector<int> t0, t1, mrow_id, mrow_ptr(1, 0), big;
    struct Seg { int beg, end, dest, row; };
    std::vector<Seg> segs;
    p.t0_cost_prefix.assign(1, 0);
    int nslots = 0;
    for (int r = 0; r < V; ++r) {
        const int beg = c->h_ptr[r], end = c->h_ptr[r + 1], deg = end - beg;
        if (deg <= chunk) {
            t0.insert(t0.end(), {beg, end, r, r});
            p.t0_cost_prefix.push_back(p.t0_cost_prefix.back() + deg + kItemCost);
        } else if (deg <= seg_edges) {
            segs.push_back({beg, end, r, r});
        } else {
            const int nseg = (int)((deg + seg_edges - 1) / seg_edges);
            for (int j = 0; j < nseg; ++j) {
                const long sb = beg + (long)j * seg_edges;
                segs.push_back({(int)sb, (int)std::min<long>(sb + seg_edges, end), ~(nslots + j), r});
            }
            if (nseg > kBigRowPartials) big.push_back((int)mrow_id.size());
            nslots += nseg;
            mrow_id.push_back(r);
            mrow_ptr.push_back(nslots);
        }
    }
    // heaviest segments first: they start at t = 0 and never form the tail of the launch
    std::stable_sort(segs.begin(), segs.end(), [](const Seg &a, const Seg &b) { return a.end - a.beg > b.end - b.beg; });
    for (const Seg &sg : segs) t1.insert(t1.end(), {sg.beg, sg.end, sg.dest, sg.row});
    p.n0 = (int)(t0.size() / 4);
    p.n1 = (int)(t1.size() / 4);
    p.n_mrows = (int)mrow_id.size();
    p.n_slots = nslots;
    p.n_big = (int)big.size();
    if ((rc = p.t0.upload(t0))) return rc;
    p.h_t0 = t0;
    if ((rc = p.t1.upload(t1))) return rc;
    if ((rc = p.mrow_id.upload(mrow_id))) return rc;
    if ((rc = p.mrow_ptr.upload(mrow_ptr))) return rc;
    if ((rc = p.big_rows.upload(big))) return rc;
    {
        std::vector<int> slot_hub((size_t)nslots);
        for (size_t m = 0; m + 1 < mrow_ptr.size(); ++m)
            for (int sl = mrow_ptr[m]; sl < mrow_ptr[m + 1]; ++sl) slot_hub[sl] = (int)m;
        if ((rc = p.slot_hub.upload(slot_hub))) return rc;
    }
    if (padding_ratio) {
        // lane groups a segment workgroup occupies (rounds x groups, taken as 8 groups) vs the chunks it really has
        long padded = 0, chunks = p.n0;
        for (const Seg &sg : segs) {
            const long nch = ((long)sg.end - sg.beg + chunk - 1) / chunk;
            padded += 8 * ((nch + 7) / 8);
            chunks += nch;
        }
        *padding_ratio = chunks > 0 ? (double)(padded + p.n0) / (double)chunks : 1.0;
    }
    if (describe_in_sched1) {
        // host description of the summation order (chunks of `chunk` edges per row) for get_schedule()
        Schedule &s = c->sched[1];
        s.reset();
        s.kind = GNNAGG_SCHED_NEIGHBOR_GROUPING;
        const int G = neighbor_grouping(c->h_ptr.data(), chunk, V, nullptr, nullptr);
        s.h_ptr_s.resize((size_t)G + 1);
        s.h_target.resize((size_t)G);
        neighbor_grouping(c->h_ptr.data(), chunk, V, s.h_ptr_s.data(), s.h_target.data());
        s.num_target = G;
    }
    p.valid = true;
    return GNNAGG_OK;
}

static int build_balanced_plan(Ctx *c, int chunk) { return build_plan_into(c, c->plan, chunk, true, nullptr); }
static int pick_chunk(const Ctx *c);
// the plan alone, leaving sched[1] (a source-partitioned schedule) untouched: GNNAGG_FLAG_ACCUMULATE on such a handle
static int build_balanced_plan_keep(Ctx *c) { return build_plan_into(c, c->plan, pick_chunk(c), false, nullptr); }

// Narrow features (8- and 16-lane groups): a wavefront holds 8 or 4 rows and runs for the longest of them, so rows of
// similar degree should share a wavefront.  Sorting by degree inside windows of `sort_window` rows equalises them while
// consecutive windows (and with them the XCD ranges) still follow the row order.  Measured on the arxiv-shaped input:
// F=32 43.3 -> 36.8 us, F=64 51.6 -> 51.0 us; F=128 (2 rows per wavefront) gets slower, so it keeps the row order.
static bool wants_sorted_rows(const Ctx *c, int feat) { return c->sort_window > 1 && feat <= 64; }

static int ensure_sorted_rows(Ctx *c, BalancedPlan &p)
{
    if (p.t0_sorted.p || p.n0 == 0) return GNNAGG_OK;
    const int n0 = p.n0;
    const std::vector<int> &t0 = p.h_t0;
    std::vector<int> order((size_t)n0);
    for (int i = 0; i < n0; ++i) order[i] = i;
    for (int w0 = 0; w0 < n0; w0 += c->sort_window) {
        const int w1 = std::min(n0, w0 + c->sort_window);
        std::stable_sort(order.begin() + w0, order.begin() + w1,
                         [&](int a, int b) { return t0[4 * a + 1] - t0[4 * a] > t0[4 * b + 1] - t0[4 * b]; });
    }
    std::vector<int> sorted(t0.size());
    p.t0s_cost_prefix.assign((size_t)n0 + 1, 0);
    for (int i = 0; i < n0; ++i) {
        for (int q = 0; q < 4; ++q) sorted[4 * i + q] = t0[4 * order[i] + q];
        p.t0s_cost_prefix[i + 1] = p.t0s_cost_prefix[i] + (sorted[4 * i + 1] - sorted[4 * i]) + kItemCost;
    }
    return p.t0_sorted.upload(sorted);
}

static int build_rows_plan(Ctx *c, bool medium_ok = true)
{
    int rc = fetch_host_ptr(c);
    if (rc) return rc;
    RowsPlan &p = medium_ok ? c->rows_plan : c->rows_plan_nomed;
    p.valid = false;
    p.medium_ok = medium_ok;
    // The workgroup-per-row kernel finishes a long chain sooner (parallel gathers, one lane per column consuming), but a CU
    // holds one such workgroup while it could hold dozens of lane groups walking their own rows, so it only takes rows far
    // above the average degree: max(1024, 16 * avg).  Measured (rows mode, ms): reddit-shaped SAGE F=602 56.3 with every
    // row on lane groups, 45.8 at 4 * avg, 41.0 at 16 * avg, 40.0 at 32 * avg; reddit-shaped GAT 47.2 / 19.8 / 16.8 / 16.5;
    // products-shaped 9.8 at 1024 but 89 at 256 (the short rows starve).
    p.long_deg = std::max(1024, 16 * c->avg_deg());
    // Between the two: a lane group walks its row 8 gathers at a time, each batch waiting for the one before it -- a 1000-edge row
    // is ~125 dependent round trips (~150 us), and on a small graph that tail IS the launch (arxiv-shaped: 170 us of short rows
    // against 74 us for the whole balanced launch).  Rows above med_deg edges take the 128-thread form of the long-row kernel
    // instead: one gather wavefront + the consumer, 16.5 KB of LDS, nine workgroups per CU -- no CU is taken away from the short
    // rows the way the 512-thread form does it.  Per edge that form is dearer than a lane group (half of its lanes consume), so it
    // only pays where the tail shows: med_deg grows with the graph (a lane group's longest row should stay ~1/5 of what the launch
    // takes anyway) and the class is empty from 4.6 M edges on.  Measured, rows mode: arxiv-shaped 186 -> 121-127 us (thresholds
    // 256 / 128; 512: 135); products-shaped 9.8 ms without the class, 10.0-11.9 ms with thresholds 512 ... 64.
    p.med_deg = c->opt_rows_medium > 0 ? c->opt_rows_medium : c->opt_rows_medium < 0 ? p.long_deg : std::max(128, (int)(c->E / 4500));
    p.med_deg = std::min(p.med_deg, p.long_deg);
    // (GAT, head width not a multiple of 32: the long-row kernel's 32-column tiles would straddle heads, so the medium class -- whose
    // only kernel that is -- stays with the lane groups; a graph without hub rows then runs entirely on k_gat_plan's descriptor path)
    if (!medium_ok) p.med_deg = p.long_deg;
    std::vector<int> r0;
    struct Long { int beg, end, row; };
    std::vector<Long> longs, meds;
    p.r0_cost_prefix.assign(1, 0);
    for (int r = 0; r < c->V; ++r) {
        const int beg = c->h_ptr[r], end = c->h_ptr[r + 1];
        if (end - beg <= p.med_deg) {
            r0.insert(r0.end(), {beg, end, r, r});
            p.r0_cost_prefix.push_back(p.r0_cost_prefix.back() + (end - beg) + kItemCost);
        } else if (end - beg <= p.long_deg) {
            meds.push_back({beg, end, r});
        } else {
            longs.push_back({beg, end, r});
        }
    }
    const auto heavier = [](const Long &a, const Long &b) { return a.end - a.beg > b.end - b.beg; };
    std::stable_sort(longs.begin(), longs.end(), heavier);
    std::stable_sort(meds.begin(), meds.end(), heavier);
    std::vector<int> r1, r2, rows;
    for (const Long &l : longs) { r1.insert(r1.end(), {l.beg, l.end, l.row, 0}); rows.push_back(l.row); }
    for (const Long &l : meds) { r2.insert(r2.end(), {l.beg, l.end, l.row, 0}); rows.push_back(l.row); }
    p.n0 = (int)(r0.size() / 4);
    p.n1 = (int)(r1.size() / 4);
    p.n2 = (int)(r2.size() / 4);
    if ((rc = p.r0.upload(r0))) return rc;
    if ((rc = p.r1.upload(r1))) return rc;
    if ((rc = p.r2.upload(r2))) return rc;
    if ((rc = p.r1_rows.upload(rows))) return rc;
    p.valid = true;
    return GNNAGG_OK;
}

static int pick_chunk(const Ctx *c)
{
    // long rows become work items of <= chunk edges: small enough that the hub rows of a
    // power-law graph spread over many wavefronts, large enough that partial-sum traffic
    // (2 * F * 4 bytes per extra item) stays a few percent of the gather traffic.
    int chunk = 64;  // measured on arxiv-shaped input: 64..96 beats 32 (fewer partial rows) and 128+ (tail)
    while (chunk < 512 && chunk < 2 * c->avg_deg()) chunk <<= 1;
    return chunk;
}

// Source-partitioned ("2-D blocked") balanced mode.  On high-degree graphs the aggregation is bound by L2-miss traffic: rows
// gathered from the Infinity Cache or HBM arrive at 6.3-7.9 TB/s, rows gathered from an XCD's own 4 MB L2 at 24 TB/s
// (256-byte segments; scripts/micro/gather_ceiling.hip, profiles/r02/gather_ceiling.txt).  The reference's locality
// schedule (graph_schedule.h:156-243: per column range, per row, the sub-row of edges whose source falls in the range, cut
// every NG edges) supplies the order; two things make the slice an L2 walks actually fit:
//  * the features are processed one COLUMN TILE of tile_w floats at a time (tile-major block order), so the slice is
//    (rows of a range) x (tile_w * 4 bytes), not (rows of a range) x (row pitch);
//  * the number of ranges is chosen from the slice size, P = ceil(columns * tile_w * 4 / slice bytes), bounded so that a
//    (row, range) sub-row keeps about a dozen edges on average (every (row, range, tile) costs one partial row).
// Per-column summation order does not depend on the tiling: partials of a row are folded in ascending group order exactly
// as the reference's arrays list them (restated by orc_locality_schedule + orc_gcn_grouped_seg with seg = 0).
// Round 1 (16 ranges x full rows, 35 MB slices): reddit-shaped SAGE F=602 36.9 -> 28.8 ms, L2 hit 0.08 -> 0.41.
// GNNAGG_PARTITIONS = 0 / N overrides; GNNAGG_TILE_W, GNNAGG_SLICE_KB tune the slice.
static int parts_for_cols(const Ctx *c, long cols)
{
    const long slice = std::max(1L, (long)c->opt_slice_kb) * 1024;
    long p = (cols * c->opt_tile_w * 4 + slice - 1) / slice;
    p = std::min<long>(p, std::max(1, c->avg_deg() / 12));
    return (int)std::max(1L, std::min(p, 1024L));
}

// 0: chunked plan; -1: source-partitioned with the range count taken from the column count; N > 0: N ranges
static int auto_partitions(const Ctx *c)
{
    if (c->no_auto_partition) return 0;
    if (c->opt_partitions >= 0) return c->opt_partitions;
    return c->avg_deg() >= c->opt_part_min_deg ? -1 : 0;
}

static int build_locality(Ctx *c, Schedule &s, int par_num, int ng, int total_v, int kind, bool keep_eid = false);

static int build_spans(Ctx *c, Schedule &s);

static constexpr int kSpanEdges = 512;

// The blocked order built on the device (plan_gpu.hip): only what the segmented-stream kernels read.  `parts` as in build_locality
// (-1: from the column count).  *done = false: not applicable here (the caller builds on the host).
static int build_partitioned_gpu(Ctx *c, int parts, int ng, bool *done)
{
    *done = false;
    if (c->E <= 0 || c->V <= 0 || c->force_host_plan) return GNNAGG_OK;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || (size_t)c->E * 72 > free_b / 2) return GNNAGG_OK;   // ~ 50 B / edge of temporaries
    int rc = fetch_host_ptr(c);
    if (rc) return rc;
    int mx = 0;
    if ((rc = gpu_max_col(c->d_idx, c->E, c->stream, &mx))) return rc;
    if ((unsigned)mx > 0x3fffffffu) return GNNAGG_OK;   // the two flag bits are not free
    int par_num = parts == -1 ? parts_for_cols(c, (long)mx + 1) : parts;
    par_num = std::max(1, std::min(par_num, mx + 1));
    if (par_num > 65535) return GNNAGG_OK;   // the device builder keeps the range in 16 bits; the host builder takes any count ("partitions" = N)
    GpuBlockedPlan g;
    if ((rc = gpu_build_blocked_plan(c->d_ptr, c->d_idx, c->V, c->E, par_num, mx + 1, ng, kSpanEdges, c->stream, g))) return rc;
    Schedule &s = c->sched[1];
    s.reset();
    s.kind = GNNAGG_SCHED_LOCALITY_NEIGHBOR_GROUPING;
    s.permuted = true; s.gpu_built = true;
    s.total_cols = g.total_cols; s.par_num = g.par_num; s.num_target = g.G; s.n_edges_perm = g.n_edges;
    s.n_slots = g.G; s.n_empty = g.n_empty; s.n_spans = g.n_spans; s.n_crows = g.n_crows;
    s.ptr_s.swap(g.ptr_s); s.target.swap(g.target); s.eperm.swap(g.eperm); s.idx_f.swap(g.idx_f); s.span_g.swap(g.span_g);
    s.rg_ptr.swap(g.rg_ptr); s.rg_idx.swap(g.rg_idx); s.crows.swap(g.crows); s.empty_rows.swap(g.empty_rows);
    s.h_ptr_s.swap(g.h_ptr_s); s.h_target.swap(g.h_target); s.h_empty.swap(g.h_empty); s.span_cost_prefix.swap(g.span_cost_prefix);
    s.valid = true;
    c->partitions = s.par_num;
    c->plan_part.chunk = ng;
    c->plan_part.valid = true;   // (no descriptors: a run that needs them rebuilds on the host, see force_host_plan)
    *done = true;
    return GNNAGG_OK;
}

static size_t sched_device_bytes(const Schedule &s)
{
    return (s.ptr_s.n + s.target.n + s.slot.n + s.empty_rows.n + s.mrow_id.n + s.mrow_ptr.n + s.idx_s.n + s.big_rows.n + s.eperm.n + s.idx_f.n +
            s.span_g.n + s.crows.n + s.rg_ptr.n + s.rg_idx.n + s.val_s.n) * sizeof(int);
}

static int build_partitioned_host(Ctx *c, int parts);

static int build_partitioned(Ctx *c, int parts)
{
    const double t0 = wall_seconds();
    c->plan.reset();
    c->plan_part.reset();
    constexpr int span_chunk_gpu = 128;
    bool done = false;
    int rc = GNNAGG_OK;
    if (c->tiled && c->use_spans && c->use_plan) rc = build_partitioned_gpu(c, parts, std::min(pick_chunk(c), span_chunk_gpu), &done);
    if (!rc && !done) rc = build_partitioned_host(c, parts);
    c->plan_seconds = wall_seconds() - t0;
    c->plan_bytes = sched_device_bytes(c->sched[1]) + c->plan_part.t0.n * sizeof(int);
    return rc;
}

static int build_partitioned_host(Ctx *c, int parts)
{
    c->plan.reset();
    c->plan_part.reset();
    Schedule &s = c->sched[1];
    // the segmented-stream kernel wants several groups per span: groups of at most 128 edges there
    constexpr int span_chunk = 128;
    const bool spans = c->tiled && c->use_spans;
    int rc = build_locality(c, s, parts, spans ? std::min(pick_chunk(c), span_chunk) : pick_chunk(c), -1,
                            GNNAGG_SCHED_LOCALITY_NEIGHBOR_GROUPING, true);
    if (rc) return rc;
    c->partitions = s.par_num;
    // the same groups as 16-byte descriptors {beg, end, dest, row} for the plan kernels' short-row path (dest < 0: ~scratch
    // slot of a row with several groups), rows without edges behind them; the order and the XCD costs are the schedule's
    BalancedPlan &p = c->plan_part;
    const int G = s.num_target;
    std::vector<int> t0;
    t0.reserve(((size_t)G + s.h_empty.size()) * 4);
    for (int g = 0; g < G; ++g) {
        const int r = s.h_target[g];
        t0.insert(t0.end(), {s.h_ptr_s[g], s.h_ptr_s[g + 1], s.h_slot[g] >= 0 ? ~s.h_slot[g] : r, r});
    }
    for (int r : s.h_empty) t0.insert(t0.end(), {0, 0, r, r});
    p.n0 = (int)(t0.size() / 4);
    p.chunk = spans ? std::min(pick_chunk(c), span_chunk) : pick_chunk(c);
    p.t0_cost_prefix = s.cost_prefix;
    if ((rc = p.t0.upload(t0))) return rc;
    p.valid = true;
    if (spans && (rc = build_spans(c, s))) return rc;
    return GNNAGG_OK;
}

// Segmented-stream form of the partitioned order: spans of whole groups with about kSpanEdges edges each, the ids with
// the group-end flags, and the row -> groups lists of the ordered combine.
static int build_spans(Ctx *c, Schedule &s)
{
    const int G = s.num_target, V = c->V;
    if (G == 0) return GNNAGG_OK;
    int mx = 0;
    for (int v : s.h_idx_s) mx = std::max(mx, v);
    if ((unsigned)mx > 0x3fffffffu) return GNNAGG_OK;  // the two flag bits are not free: stay on the descriptor kernels
    const int span_edges = kSpanEdges;
    std::vector<int> groups_of((size_t)V, 0);
    for (int g = 0; g < G; ++g) groups_of[s.h_target[g]]++;
    // flagged ids
    std::vector<int> idx_f(s.h_idx_s);
    for (int g = 0; g < G; ++g) {
        const int last = s.h_ptr_s[g + 1] - 1;
        unsigned w = (unsigned)idx_f[last] | 0x80000000u;
        if (groups_of[s.h_target[g]] == 1) w |= 0x40000000u;
        idx_f[last] = (int)w;
    }
    // spans
    std::vector<int> span_g(1, 0);
    s.span_cost_prefix.assign(1, 0);
    for (int g = 0; g < G;) {
        const int e0 = s.h_ptr_s[g];
        int h = g + 1;
        while (h < G && s.h_ptr_s[h] - e0 < span_edges) ++h;
        span_g.push_back(h);
        s.span_cost_prefix.push_back((long)s.h_ptr_s[h]);
        g = h;
    }
    s.n_spans = (int)span_g.size() - 1;
    // row -> groups (stable counting sort keeps the ascending group order), rows with several groups heaviest first
    std::vector<int> rg_ptr((size_t)V + 1, 0), rg_idx((size_t)G);
    for (int r = 0; r < V; ++r) rg_ptr[r + 1] = rg_ptr[r] + groups_of[r];
    {
        std::vector<int> cur(rg_ptr.begin(), rg_ptr.end() - 1);
        for (int g = 0; g < G; ++g) rg_idx[cur[s.h_target[g]]++] = g;
    }
    std::vector<int> crows;
    for (int r = 0; r < V; ++r)
        if (groups_of[r] > 1) crows.push_back(r);
    std::stable_sort(crows.begin(), crows.end(), [&](int a, int b) { return groups_of[a] > groups_of[b]; });
    s.n_crows = (int)crows.size();
    int rc;
    if ((rc = s.idx_f.upload(idx_f)) || (rc = s.span_g.upload(span_g)) || (rc = s.crows.upload(crows)) ||
        (rc = s.rg_ptr.upload(rg_ptr)) || (rc = s.rg_idx.upload(rg_idx)))
        return rc;
    return GNNAGG_OK;
}

// The source-partitioned order was chosen by the library, not by the caller, so the aliasing contract of updateval
// (aggr_gcn.h:540-544: the aggregator reads the caller's array at run time) has to survive the permutation: the permuted
// copy of the edge values is re-gathered from the caller's array before every run (E floats; 0.13 ms at 115 M edges).
static int refresh_partitioned_val(Ctx *c, Schedule *s)
{
    if (c->kind != Ctx::GCN || !c->d_val || !s->eperm.p) return GNNAGG_OK;
    const size_t kept = s->eperm.n;
    int rc = s->val_s.reserve(kept);
    if (rc) return rc;
    return launch_permute_val(s->eperm.p, c->d_val, s->val_s.p, (int)kept, c->stream);
}

static int get_sched(Ctx *c, int mode, Schedule **out);

// Moves a handle from the source-partitioned order to the chunked plan for good (its scratch does not fit):
// gnnagg_balanced_partitions reports 0 from then on and sched[1] describes the chunked order again.
static int demote_partitioned(Ctx *c)
{
    c->partitions = 0;
    c->no_auto_partition = 1;
    c->sched[1].reset();
    c->plan_part.reset();
    c->xt.release();
    return build_balanced_plan(c, pick_chunk(c));
}

// Scratch of a run on the partitioned order: one partial row per (row with several groups, group) -- n_slots * padded
// feature length floats (11 GB for the reddit-shaped F = 602 case) -- plus the tiled image of X.  Grown on demand; when the
// growth would take more than half of the memory that is free right now, or the allocation fails, *demoted is set and the
// caller re-dispatches on the chunked plan.
static int reserve_partitioned_scratch(Ctx *c, size_t partial_floats, size_t den_floats, size_t xt_floats, bool *demoted)
{
    *demoted = false;
    const size_t grow = (partial_floats > c->partial.n ? partial_floats : 0) + (den_floats > c->partial_den.n ? den_floats : 0) +
                        (xt_floats > c->xt.n ? xt_floats : 0);
    if (grow == 0) return GNNAGG_OK;  // no API call: capture-safe once warm
    size_t free_b = 0, total_b = 0;
    bool ok = hipMemGetInfo(&free_b, &total_b) == hipSuccess && grow * sizeof(float) <= free_b / 2 + (c->partial.n + c->partial_den.n + c->xt.n) * sizeof(float);
    if (c->opt_scratch_limit_mb > 0 && (partial_floats + den_floats + xt_floats) * sizeof(float) > (size_t)c->opt_scratch_limit_mb << 20) ok = false;
    if (ok) {
        ok = c->partial.reserve(partial_floats) == GNNAGG_OK && c->partial_den.reserve(den_floats) == GNNAGG_OK &&
             c->xt.reserve(xt_floats) == GNNAGG_OK;
        if (!ok) (void)hipGetLastError();  // out of memory is handled here, not reported
    }
    if (ok) return GNNAGG_OK;
    *demoted = true;
    return demote_partitioned(c);
}

// TileSpec of one run on the partitioned order: lane geometry, strides, and which image of X the kernel reads.
struct TiledRun {
    TileSpec spec;
    int ntiles = 1;
    bool retile = false;
    size_t partial_floats = 0, xt_floats = 0;
};

static TiledRun plan_tiles(const Ctx *c, const Schedule &s, const float *x, const float *y, int feat, int lane_unit)
{
    TiledRun t;
    t.partial_floats = (size_t)s.n_slots * feat;
    if (!c->tiled || lane_unit % 4 != 0) return t;  // (GAT: a lane's 4 columns must belong to one head)
    int tw = c->opt_tile_w;
    while (tw > 32 && tw / 2 >= feat) tw /= 2;  // narrow features: no lanes on columns that do not exist (F <= 32: 8-lane groups)
    t.ntiles = (feat + tw - 1) / tw;
    const bool direct_ok = ((size_t)feat * 4) % 128 == 0 && ((uintptr_t)x % 128) == 0;
    t.retile = c->opt_retile != 0 || !direct_ok;
    t.spec.on = 1;
    t.spec.tile_w = tw;
    t.spec.xpitch = t.retile ? tw : feat;
    t.spec.x_tile_stride = t.retile ? (long)s.total_cols * tw : tw;
    t.spec.ppitch = tw;
    t.spec.p_tile_stride = (long)s.n_slots * tw;
    t.spec.yvec = (feat % 4 == 0 && (uintptr_t)y % 16 == 0) ? 4 : (feat % 2 == 0 && (uintptr_t)y % 8 == 0) ? 2 : 1;
    t.partial_floats = (size_t)s.n_slots * tw * t.ntiles;
    t.xt_floats = t.retile ? (size_t)s.total_cols * tw * t.ntiles : 0;
    return t;
}

static int get_sched(Ctx *c, int mode, Schedule **out)
{
    // "fast_rows": the reference drivers' run(vin, vout, B, 0) gets the balanced order (gnnagg_set_option)
    if (mode == GNNAGG_MODE_ROWS && c->fast_rows) mode = GNNAGG_MODE_BALANCED;
    if (mode == GNNAGG_MODE_SCHEDULED) {
        if (!c->sched[0].valid)
            return fail(GNNAGG_ERR_STATE, "scheduled run without schedule() (reference: assert aggr_gcn.h:392)");
        *out = &c->sched[0];
    } else if (mode == GNNAGG_MODE_BALANCED) {
        if (c->partitions > 0 || (c->use_plan && !c->plan.valid && auto_partitions(c) != 0)) {
            if (!c->sched[1].valid) {
                int rc = build_partitioned(c, c->partitions > 0 ? c->partitions : auto_partitions(c));
                if (rc) return rc;
            }
        } else if (c->use_plan) {
            if (!c->plan.valid) {
                int rc = build_balanced_plan(c, pick_chunk(c));
                if (rc) return rc;
            }
        } else if (!c->sched[1].valid) {
            int rc = build_grouping(c, c->sched[1], pick_chunk(c), GNNAGG_SCHED_NEIGHBOR_GROUPING);
            if (rc) return rc;
        }
        *out = &c->sched[1];
    } else {
        *out = nullptr;
    }
    return GNNAGG_OK;
}

// Arrival counters of the in-kernel hub fold: zero whenever no launch is in flight (the last arriver resets its own).
static int reserve_hub_counters(Ctx *c, int n_mrows, int feat, int *stride_out)
{
    const int stride = feat / 64 + 2;  // >= column tiles of any lane geometry
    const size_t want = (size_t)n_mrows * stride;
    if (c->hub_count.n < want) {
        int rc = c->hub_count.reserve(want);
        if (rc) return rc;
        // stream-ordered with the launch that uses them: a null-stream hipMemset is NOT ordered against a caller's non-blocking stream
        // (round 6: one first step in ~12 folded its hubs on counters that were not zero yet)
        HIP_TRY(hipMemsetAsync(c->hub_count.p, 0, c->hub_count.n * sizeof(int), c->stream));
    }
    *stride_out = stride;
    return GNNAGG_OK;
}


// Canonical rows mode (GNNAGG_MODE_ROWS: one sequential chain per (row, column) in CSR order, aggr_gcn.h:13-35) on the 2-D blocked
// order.  If every row lists its neighbors in ascending order, the row's sub-rows per source range, taken range after range, ARE the
// row in CSR order: the reference's locality_schedule arrays (graph_schedule.h:17-63, one group per (row, range), no neighbor
// grouping) walked one range per launch, every group's chain starting from what the row's earlier ranges left in a tiled image of Y
// and returning there (k_gcn_span<..., CHAIN>), give exactly the canonical chains -- with the gathers served by the L2 instead of
// the fabric (reddit-shaped SAGE F = 602: 41 ms on the row kernels).  Applies to graphs the balanced mode would block as well
// (average degree >= partition_min_degree), sum / mean, 64-float tiles; everything else stays on the row kernels.
static int build_rows_blocked_host(Ctx *c, int ntiles_hint);

// the chain plan on the device (plan_gpu.hip); *done = false: not applicable (the host builder decides)
static int build_rows_blocked_gpu(Ctx *c, int ntiles_hint, bool *done)
{
    *done = false;
    Ctx::RowsBlocked &rb = c->rb;
    size_t free_b = 0, total_b = 0;
    if (c->force_host_plan || hipMemGetInfo(&free_b, &total_b) != hipSuccess || (size_t)c->E * 80 > free_b / 2) return GNNAGG_OK;
    int mx = 0, rc;
    if ((rc = gpu_max_col(c->d_idx, c->E, c->stream, &mx))) return rc;
    if (mx + 1 >= (1 << 24)) { *done = true; return GNNAGG_OK; }   // (24-bit ids in the chained kernel: the row kernels keep such graphs)
    const int slice_kb = c->opt_slice_kb;
    if (ntiles_hint <= 4 && c->opt_partitions < 0) c->opt_slice_kb = 2 * slice_kb;   // see build_rows_blocked_host
    int par_num = c->opt_partitions > 0 ? c->opt_partitions : parts_for_cols(c, (long)mx + 1);
    c->opt_slice_kb = slice_kb;
    par_num = std::max(1, std::min(par_num, mx + 1));
    *done = true;
    if (par_num < 2) return GNNAGG_OK;
    // (GAT: the workgroup-per-row kernel pays an exp per edge and 32-column tile -- 8.9 ms beside 9.5 ms of chained launches on the
    // reddit-shaped 8 x 32 case at 1024 -- so only sub-rows a lane group cannot finish inside a launch leave the chains:
    // 12.05 / 10.89 / 10.40 / 10.23 ms at 1024 / 2048 / 4096 / 8192, 13.65 ms without hub rows; profiles/r04/rows_gat_sweep.txt)
    const int hub_edges = c->opt_rb_hub_edges > 0 ? c->opt_rb_hub_edges
                          : c->kind == Ctx::GAT ? 8192 : std::min(4096, std::max(512, 256 * std::max(1, ntiles_hint)));
    GpuChainPlan g;
    if ((rc = gpu_build_chain_plan(c->d_ptr, c->d_idx, c->h_ptr.data(), c->V, c->E, par_num, mx + 1, hub_edges, kSpanEdges, c->stream, g))) return rc;
    if (!g.sorted_rows || g.G == 0) return GNNAGG_OK;
    Schedule &s = rb.sched;
    s.reset();
    s.kind = GNNAGG_SCHED_LOCALITY; s.permuted = true; s.gpu_built = true;
    s.total_cols = g.total_cols; s.par_num = g.par_num; s.num_target = g.G; s.n_edges_perm = g.n_edges; s.n_slots = 0;
    s.ptr_s.swap(g.ptr_s); s.target.swap(g.target); s.eperm.swap(g.eperm);
    s.h_ptr_s.swap(g.h_ptr_s); s.h_target.swap(g.h_target);
    s.valid = true;
    rb.idx_f.swap(g.idx_f); rb.span_g.swap(g.span_g); rb.r1.swap(g.r1); rb.hub_mask.swap(g.hub_mask);
    rb.n1 = g.n_hub; rb.span0.swap(g.span0); rb.cost.swap(g.cost);
    rb.ok = true;
    return GNNAGG_OK;
}

static int build_rows_blocked(Ctx *c, int ntiles_hint)
{
    const double t0 = wall_seconds();
    Ctx::RowsBlocked &rb = c->rb;
    rb.reset();
    rb.tried = true;
    int rc = fetch_host_ptr(c);
    if (rc) return rc;
    if (c->E == 0 || c->no_auto_partition || (c->opt_partitions < 0 && c->avg_deg() < c->opt_part_min_deg) || c->opt_partitions == 0) return GNNAGG_OK;
    bool done = false;
    rc = build_rows_blocked_gpu(c, ntiles_hint, &done);
    if (!rc && !done) rc = build_rows_blocked_host(c, ntiles_hint);
    c->rb_plan_seconds = wall_seconds() - t0;
    return rc;
}

static int build_rows_blocked_host(Ctx *c, int ntiles_hint)
{
    Ctx::RowsBlocked &rb = c->rb;
    int rc;
    {   // neighbors ascending in every row?
        std::vector<int> h_idx((size_t)c->E);
        if (int rc_ = copy_to_host(c, h_idx.data(), c->d_idx, (size_t)c->E * sizeof(int))) return rc_;
        int unsorted = 0;
#pragma omp parallel for schedule(static) reduction(| : unsorted)
        for (int r = 0; r < c->V; ++r)
            for (int e = c->h_ptr[r] + 1; e < c->h_ptr[r + 1]; ++e) unsorted |= h_idx[e] < h_idx[e - 1];
        if (unsorted) return GNNAGG_OK;
    }
    Schedule &s = rb.sched;
    // Every range is a launch here, so narrow feature widths (short launches) take slices of twice the size: reddit-shaped, 8 MB
    // instead of 4 MB: F = 128 6.5 -> 5.2 ms, F = 256 10.3 -> 9.2 ms; F = 602 (10 tiles) 20.0 -> 20.1 ms: unchanged there
    const int slice_kb = c->opt_slice_kb;
    if (ntiles_hint <= 4 && c->opt_partitions < 0) c->opt_slice_kb = 2 * slice_kb;
    c->keep_h_eperm = true;
    rc = build_locality(c, s, c->opt_partitions > 0 ? c->opt_partitions : -1, 0, -1, GNNAGG_SCHED_LOCALITY, true);
    c->keep_h_eperm = false;
    c->opt_slice_kb = slice_kb;
    if (rc) return rc;
    if (s.num_target == 0 || s.par_num < 2 || s.total_cols >= (1 << 24)) return GNNAGG_OK;
    {   // A sub-row is walked by ONE lane group, a few edges per microsecond: rows with a sub-row of more than hub_edges edges would
        // set the duration of their range's launch.  They go to the workgroup-per-row kernel of the rows mode (k_gcn_rows_long:
        // parallel gathers, one consuming wavefront; the same canonical chain) after the other rows' chains, whole.
        // The threshold follows the duration of a launch, i.e. the number of column tiles of the first run that builds the plan
        // (reddit-shaped, ms per step at 512 / 1024 / 2048 / 4096 edges: F = 128 5.6 / 5.7 / 7.1 / 7.5, F = 256 10.5 / 10.1 / 10.3 /
        // 10.3, F = 602 -- 10 tiles -- 23.9 / 20.2 / 19.7 at 1024 / 2048 / 4096).
        const int hub_edges = c->opt_rb_hub_edges > 0 ? c->opt_rb_hub_edges
                              : c->kind == Ctx::GAT ? 8192 : std::min(4096, std::max(512, 256 * std::max(1, ntiles_hint)));
        const int G0 = s.num_target;
        std::vector<char> is_hub((size_t)c->V, 0);
        int n_hub = 0;
        for (int g = 0; g < G0; ++g)
            if (s.h_ptr_s[g + 1] - s.h_ptr_s[g] > hub_edges && !is_hub[s.h_target[g]]) { is_hub[s.h_target[g]] = 1; ++n_hub; }
        {   // the groups that stay, range by range, the longest sub-rows of a range first (they are walked by one lane group each: started
            // first they do not form the tail of their launch); any order of the rows inside a range gives the same chains
            const int P0 = s.par_num, w0 = s.total_cols / P0;
            auto part0 = [&](int col) { const int p = col / w0; return p >= P0 ? P0 - 1 : p; };
            std::vector<int> order;
            order.reserve((size_t)G0);
            for (int g = 0; g < G0; ++g)
                if (!is_hub[s.h_target[g]]) order.push_back(g);
            std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
                const int pa = part0(s.h_idx_s[s.h_ptr_s[a]]), pb = part0(s.h_idx_s[s.h_ptr_s[b]]);
                if (pa != pb) return pa < pb;
                return s.h_ptr_s[a + 1] - s.h_ptr_s[a] > s.h_ptr_s[b + 1] - s.h_ptr_s[b];
            });
            std::vector<int> np(1, 0), nt, ni, ne;
            nt.reserve(order.size()); ni.reserve(s.h_idx_s.size()); ne.reserve(s.h_idx_s.size());
            for (int g : order) {
                nt.push_back(s.h_target[g]);
                ni.insert(ni.end(), s.h_idx_s.begin() + s.h_ptr_s[g], s.h_idx_s.begin() + s.h_ptr_s[g + 1]);
                ne.insert(ne.end(), s.h_eperm.begin() + s.h_ptr_s[g], s.h_eperm.begin() + s.h_ptr_s[g + 1]);
                np.push_back((int)ni.size());
            }
            const int par = s.par_num, cols = s.total_cols, kind = s.kind;
            s.reset();
            s.par_num = par; s.total_cols = cols; s.kind = kind; s.permuted = true;
            s.h_ptr_s.swap(np); s.h_target.swap(nt); s.h_idx_s.swap(ni); s.h_eperm.swap(ne);
            if ((rc = finalize_schedule(c, s)) || (rc = s.eperm.upload(s.h_eperm))) return rc;
        }
        if (n_hub > 0) {
            struct Long { int beg, end, row; };
            std::vector<Long> longs;
            for (int r = 0; r < c->V; ++r)
                if (is_hub[r]) longs.push_back({c->h_ptr[r], c->h_ptr[r + 1], r});
            std::stable_sort(longs.begin(), longs.end(), [](const Long &a, const Long &b) { return a.end - a.beg > b.end - b.beg; });
            std::vector<int> r1;
            for (const Long &l : longs) r1.insert(r1.end(), {l.beg, l.end, l.row, 0});
            if ((rc = rb.r1.upload(r1))) return rc;
            std::vector<unsigned char> mask(is_hub.begin(), is_hub.end());
            if ((rc = rb.hub_mask.upload(mask))) return rc;
            rb.n1 = n_hub;
        }
        s.h_eperm.clear();
        s.h_eperm.shrink_to_fit();
    }
    const int G = s.num_target, P = s.par_num;
    if (G == 0) return GNNAGG_OK;
    const int width = s.total_cols / P;
    auto part_of = [&](int col) { const int p = col / width; return p >= P ? P - 1 : p; };
    std::vector<int> idx_f(s.h_idx_s);
    for (int g = 0; g < G; ++g) idx_f[s.h_ptr_s[g + 1] - 1] = (int)((unsigned)idx_f[s.h_ptr_s[g + 1] - 1] | 0x80000000u);
    std::vector<int> span_g(1, 0);
    rb.span0.assign(1, 0);
    rb.cost.assign((size_t)P, std::vector<long>());
    int g = 0;
    for (int p = 0; p < P; ++p) {
        std::vector<long> &cost = rb.cost[(size_t)p];
        cost.assign(1, 0);
        const long base = g < G ? s.h_ptr_s[g] : 0;
        while (g < G && part_of(s.h_idx_s[s.h_ptr_s[g]]) == p) {
            const int e0 = s.h_ptr_s[g];
            int h = g + 1;
            while (h < G && s.h_ptr_s[h] - e0 < kSpanEdges && part_of(s.h_idx_s[s.h_ptr_s[h]]) == p) ++h;
            span_g.push_back(h);
            cost.push_back((long)s.h_ptr_s[h] - base);
            g = h;
        }
        rb.span0.push_back((int)span_g.size() - 1);
    }
    if (g != G) return GNNAGG_OK;   // (a source outside every range: not this path)
    if ((rc = rb.idx_f.upload(idx_f)) || (rc = rb.span_g.upload(span_g))) return rc;
    rb.ok = true;
    return GNNAGG_OK;
}

static int run_rows_blocked(Ctx *c, const float *x, float *y, int feat, int reduce, int flags, const NnRequest *nn, bool *used)
{
    *used = false;
    Ctx::RowsBlocked &rb = c->rb;
    int rc;
    if (!rb.tried && (rc = build_rows_blocked(c, (feat + 63) / 64))) return rc;
    if (!rb.ok) return GNNAGG_OK;
    Schedule &s = rb.sched;
    TiledRun tr = plan_tiles(c, s, x, y, feat, 4);
    if (!tr.spec.on || tr.spec.tile_w != 64 || (size_t)s.total_cols * tr.spec.xpitch * sizeof(float) >= 0xffffffffULL) return GNNAGG_OK;
    const size_t yt_floats = (size_t)c->V * tr.spec.tile_w * tr.ntiles;
    if ((size_t)c->V * tr.spec.tile_w * sizeof(float) >= 0x7fffffffULL) return GNNAGG_OK;
    if (yt_floats > c->yt.n || tr.xt_floats > c->xt.n) {   // first use (or a wider feature matrix): the scratch, or not this path
        size_t free_b = 0, total_b = 0;
        const size_t want = (yt_floats + tr.xt_floats) * sizeof(float);
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || want > free_b / 2 || c->yt.reserve(yt_floats) != GNNAGG_OK ||
            c->xt.reserve(tr.xt_floats) != GNNAGG_OK) {
            (void)hipGetLastError();
            rb.ok = false;
            return GNNAGG_OK;
        }
    }
    if ((rc = refresh_partitioned_val(c, &s))) return rc;
    // the rows left out of the chained launches, whole, on the workgroup-per-row kernel: beside the launches on the auxiliary stream
    // (disjoint rows of y), or behind them on this stream ("aux_stream" = 0)
    const bool fork = rb.n1 > 0 && c->use_aux_stream;
    auto hub_rows = [&](hipStream_t st) -> int {
        GcnRowsLongLaunch R;
            R.tile_w = c->opt_hub_tile;
        R.r1 = rb.r1.p; R.n1 = rb.n1; R.idx = c->d_idx; R.val = c->d_val; R.x = x; R.y = y; R.feat = feat; R.reduce = reduce;
        R.relu = (flags & GNNAGG_FLAG_RELU) ? 1 : 0;
        return launch_gcn_rows_long(R, st);
    };
    if (fork) {
        if (!c->aux_stream) {
            HIP_TRY(hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
        }
        HIP_TRY(hipEventRecord(c->ev_fork, c->stream));
        HIP_TRY(hipStreamWaitEvent(c->aux_stream, c->ev_fork, 0));
        if ((rc = hub_rows(c->aux_stream))) return rc;
        HIP_TRY(hipEventRecord(c->ev_join, c->aux_stream));
    }
    const float *xin = x;
    if (tr.retile) {
        if ((rc = launch_tile_x(x, c->xt.p, s.total_cols, feat, tr.spec.tile_w, c->stream))) return rc;
        xin = c->xt.p;
    }
    if ((rc = launch_zero_words(c->yt.p, yt_floats, c->stream))) return rc;
    SpanLaunch S;
    S.chain = 1;
    S.ptr_s = s.ptr_s.p; S.idx_f = rb.idx_f.p; S.val_s = c->d_val ? s.val_s.p : nullptr; S.target = s.target.p;
    S.n_groups = c->V;   // rows of the Yt image (the launcher sizes the partial window from it)
    S.row_ptr = c->d_ptr; S.x = xin; S.x_rows = s.total_cols; S.y = y; S.partial = c->yt.p; S.feat = feat; S.reduce = GNNAGG_REDUCE_SUM;
    S.tile = tr.spec;
    S.tile.p_tile_stride = (long)c->V * tr.spec.tile_w;
    for (int p = 0; p < s.par_num; ++p) {
        const int s0 = rb.span0[(size_t)p], s1 = rb.span0[(size_t)p + 1];
        if (s1 == s0) continue;
        S.span_g = rb.span_g.p + s0; S.n_spans = s1 - s0; S.span_cost_prefix = rb.cost[(size_t)p].data();
        if ((rc = launch_gcn_span(S, c->stream))) return rc;
    }
    if ((rc = launch_untile_y(c->yt.p, y, c->d_ptr, rb.n1 > 0 ? rb.hub_mask.p : nullptr, c->V, feat, tr.spec.tile_w, reduce == GNNAGG_REDUCE_MEAN,
                              (flags & GNNAGG_FLAG_RELU) ? 1 : 0, c->stream)))
        return rc;
    if (fork) HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_join, 0));
    else if (rb.n1 > 0 && (rc = hub_rows(c->stream))) return rc;
    *used = true;
    if (nn) return launch_dense_nn(y, nn->weight, nn->out, c->V, nn->cols, feat, c->stream);
    return GNNAGG_OK;
}

// "fast_scheduled" may replace the user's groups by the balanced order only when both cover the same edges: a locality schedule cut
// with total_num_v below the largest column id + 1 DROPS the edges beyond it (graph_schedule.h:23-44 keeps idx < total_num_v only),
// and the balanced order would put them back.
static bool sched_keeps_every_edge(const Ctx *c)
{
    const Schedule &s = c->sched[0];
    return !s.permuted || (long)s.h_idx_s.size() == (long)c->E;
}

// The same for GAT (`scheduled = 0` = aggr_gat, aggr_gat.h:116-164: one numerator chain per (row, column) and one denominator chain
// per (row, head) in CSR order, one division): k_gat_span<..., CHAIN> carries both through tiled images from range to range,
// k_untile_y divides.  Rows with a sub-row too long for one lane group go whole to the workgroup-per-row kernel's GAT flavour (head
// width % 32 == 0); graphs where that does not hold, head widths the span kernel does not tile, and callers that ask for newval stay
// on the row kernels.
static int run_rows_blocked_gat(Ctx *c, const float *x, const float *att, float *y, int feat, int heads, float slope, bool *used)
{
    *used = false;
    if (heads <= 0 || feat % heads != 0) return GNNAGG_OK;
    const int dhead = feat / heads;
    Ctx::RowsBlocked &rb = c->rb;
    int rc;
    if (!rb.tried && (rc = build_rows_blocked(c, (feat + 63) / 64))) return rc;
    if (!rb.ok) return GNNAGG_OK;
    Schedule &s = rb.sched;
    TiledRun tr = plan_tiles(c, s, x, y, feat, dhead);
    if (!tr.spec.on || tr.spec.tile_w != 64 || !gat_span_tiles(feat, heads, 64) || (rb.n1 > 0 && (dhead % 32) != 0) ||
        (size_t)s.total_cols * tr.spec.xpitch * sizeof(float) >= 0xffffffffULL || (size_t)c->V * tr.spec.tile_w * sizeof(float) >= 0x7fffffffULL)
        return GNNAGG_OK;
    const int ht = 64 >= dhead ? 64 / dhead : 1;
    const int n_hg = (heads + ht - 1) / ht, arows = c->V > s.total_cols ? c->V : s.total_cols;
    const size_t yt_floats = (size_t)c->V * tr.spec.tile_w * tr.ntiles, den_floats = (size_t)c->V * ht * tr.ntiles, half = (size_t)n_hg * arows * ht;
    if (yt_floats > c->yt.n || tr.xt_floats > c->xt.n || den_floats > c->den_t.n || 2 * half > c->att_t.n) {   // first use: the scratch, or not this path
        size_t free_b = 0, total_b = 0;
        const size_t want = (yt_floats + tr.xt_floats + den_floats + 2 * half) * sizeof(float);
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || want > free_b / 2 || c->yt.reserve(yt_floats) != GNNAGG_OK ||
            c->xt.reserve(tr.xt_floats) != GNNAGG_OK || c->den_t.reserve(den_floats) != GNNAGG_OK || c->att_t.reserve(2 * half) != GNNAGG_OK) {
            (void)hipGetLastError();
            rb.ok = false;
            return GNNAGG_OK;
        }
    }
    const bool fork = rb.n1 > 0 && c->use_aux_stream;
    auto hub_rows = [&](hipStream_t st) -> int {
        GcnRowsLongLaunch R;
            R.tile_w = c->opt_hub_tile;
        R.r1 = rb.r1.p; R.n1 = rb.n1; R.idx = c->d_idx; R.x = x; R.y = y; R.feat = feat; R.att = att; R.heads = heads; R.slope = slope;
        return launch_gcn_rows_long(R, st);
    };
    if (fork) {
        if (!c->aux_stream) {
            HIP_TRY(hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
        }
        HIP_TRY(hipEventRecord(c->ev_fork, c->stream));
        HIP_TRY(hipStreamWaitEvent(c->aux_stream, c->ev_fork, 0));
        if ((rc = hub_rows(c->aux_stream))) return rc;
        HIP_TRY(hipEventRecord(c->ev_join, c->aux_stream));
    }
    GatSpanLaunch G;
    SpanLaunch &S = G.s;
    S.chain = 1;
    S.x = x;
    if (tr.retile) {
        if ((rc = launch_tile_x(x, c->xt.p, s.total_cols, feat, tr.spec.tile_w, c->stream))) return rc;
        S.x = c->xt.p;
    }
    if ((rc = launch_tile_att(att, c->att_t.p, c->att_t.p + half, arows, heads, ht, c->stream))) return rc;
    if ((rc = launch_zero_words(c->yt.p, yt_floats, c->stream)) || (rc = launch_zero_words(c->den_t.p, den_floats, c->stream))) return rc;
    S.ptr_s = s.ptr_s.p; S.idx_f = rb.idx_f.p; S.target = s.target.p;
    S.n_groups = c->V;   // rows of the Yt image
    S.row_ptr = c->d_ptr; S.x_rows = s.total_cols; S.y = y; S.partial = c->yt.p; S.feat = feat;
    S.tile = tr.spec;
    S.tile.p_tile_stride = (long)c->V * tr.spec.tile_w;
    G.att = att; G.as_t = c->att_t.p; G.ac_t = c->att_t.p + half; G.att_rows = arows; G.heads = heads; G.slope = slope;
    G.den_t = c->den_t.p; G.den_rows = c->V;
    for (int p = 0; p < s.par_num; ++p) {
        const int s0 = rb.span0[(size_t)p], s1 = rb.span0[(size_t)p + 1];
        if (s1 == s0) continue;
        S.span_g = rb.span_g.p + s0; S.n_spans = s1 - s0; S.span_cost_prefix = rb.cost[(size_t)p].data();
        if ((rc = launch_gat_span(G, c->stream))) return rc;
    }
    if ((rc = launch_untile_y_gat(c->yt.p, c->den_t.p, y, rb.n1 > 0 ? rb.hub_mask.p : nullptr, c->V, feat, tr.spec.tile_w, ht, dhead, c->stream))) return rc;
    if (fork) HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_join, 0));
    else if (rb.n1 > 0 && (rc = hub_rows(c->stream))) return rc;
    *used = true;
    return GNNAGG_OK;
}

int gcn_run(Ctx *c, const float *x, float *y, int feat, int mode, int reduce, int flags, const NnRequest *nn, int probe)
{
    if ((flags & GNNAGG_FLAG_ACCUMULATE) && (mode != GNNAGG_MODE_BALANCED || (reduce != GNNAGG_REDUCE_SUM && !c->row_aux) || !c->use_plan))
        return fail(GNNAGG_ERR_ARG, "GNNAGG_FLAG_ACCUMULATE needs GNNAGG_MODE_BALANCED and GNNAGG_REDUCE_SUM (mean / max: gnnagg_set_row_aux first)");
    // a row_aux array changes what mean / max compute (the row-partitioned step's two passes): the chunked plan kernel implements it
    const bool aux_run = c->row_aux != nullptr && reduce != GNNAGG_REDUCE_SUM;
    if (aux_run && (mode != GNNAGG_MODE_BALANCED || !c->use_plan || nn || probe))
        return fail(GNNAGG_ERR_ARG, "gnnagg_set_row_aux applies to plain GNNAGG_MODE_BALANCED runs only");
    if (c->kind != Ctx::GCN) return fail(GNNAGG_ERR_ARG, "handle is not a GCN aggregator");
    if ((flags & GNNAGG_FLAG_RELU) && nn) return fail(GNNAGG_ERR_ARG, "GNNAGG_FLAG_RELU is not available in run_with_nn");
    if (flags & ~(GNNAGG_FLAG_ACCUMULATE | GNNAGG_FLAG_RELU)) return fail(GNNAGG_ERR_ARG, "unknown flag bits");
    if (!x || !y) return fail(GNNAGG_ERR_ARG, "null feature pointer");
    if (reduce < GNNAGG_REDUCE_SUM || reduce > GNNAGG_REDUCE_MAX) return fail(GNNAGG_ERR_ARG, "bad reduce");
    if (mode < GNNAGG_MODE_ROWS || mode > GNNAGG_MODE_BALANCED) return fail(GNNAGG_ERR_ARG, "bad mode");
    if (mode == GNNAGG_MODE_ROWS && c->fast_rows) mode = GNNAGG_MODE_BALANCED;
    if (mode == GNNAGG_MODE_SCHEDULED && c->fast_scheduled && c->sched[0].valid && sched_keeps_every_edge(c)) mode = GNNAGG_MODE_BALANCED;
    if (mode == GNNAGG_MODE_ROWS && c->opt_rows_blocked && c->tiled && c->use_plan && reduce != GNNAGG_REDUCE_MAX && !probe) {
        bool used = false;   // canonical chains on the blocked order where the graph allows it (sorted rows, high degree)
        const int rcb = run_rows_blocked(c, x, y, feat, reduce, flags, nn, &used);
        if (rcb || used) return rcb;
    }
    Schedule *s = nullptr;
    int rc = get_sched(c, mode, &s);
    if (rc) return rc;
    const bool acc_on_partitioned = ((flags & GNNAGG_FLAG_ACCUMULATE) || aux_run) && c->partitions > 0;
    if (acc_on_partitioned && !c->plan.valid && (rc = build_balanced_plan_keep(c))) return rc;  // y += A.x needs the plan kernel
    if ((mode == GNNAGG_MODE_BALANCED && c->use_plan && (c->partitions == 0 || acc_on_partitioned)) ||
        (mode == GNNAGG_MODE_SCHEDULED && c->plan_sched.valid)) {
        BalancedPlan &p = mode == GNNAGG_MODE_BALANCED ? c->plan : c->plan_sched;
        GcnPlanLaunch P;
        P.t0 = p.t0.p; P.t1 = p.t1.p; P.n0 = p.n0; P.n1 = p.n1; P.chunk = p.chunk;
        P.t0_cost_prefix = p.t0_cost_prefix.data();
        if (wants_sorted_rows(c, feat)) {
            if ((rc = ensure_sorted_rows(c, p))) return rc;
            if (p.t0_sorted.p) { P.t0 = p.t0_sorted.p; P.t0_cost_prefix = p.t0s_cost_prefix.data(); }
        }
        P.hubs.mrow_id = p.mrow_id.p; P.hubs.mrow_ptr = p.mrow_ptr.p; P.hubs.n_mrows = p.n_mrows;
        P.hubs.n_slots = p.n_slots; P.hubs.big_rows = p.big_rows.p; P.hubs.n_big = p.n_big;
        P.row_ptr = c->d_ptr; P.idx = c->d_idx; P.val = c->d_val; P.x = x; P.y = y; P.feat = feat; P.reduce = reduce;
        P.xcd_remap = c->xcd_remap; P.accumulate = (flags & GNNAGG_FLAG_ACCUMULATE) ? 1 : 0; P.relu = (flags & GNNAGG_FLAG_RELU) ? 1 : 0;
        P.row_aux = aux_run ? c->row_aux : nullptr;
        P.num_rows = c->V;
        if (p.n_slots > 0) {
            if ((rc = c->partial.reserve((size_t)p.n_slots * feat))) return rc;
            P.partial = c->partial.p;
        }
        if (nn) {
            P.nn_weight = nn->weight; P.nn_out = nn->out; P.nn_cols = nn->cols;
        }
        if (p.n_mrows > 0 && c->inkernel_combine) {
            if ((rc = reserve_hub_counters(c, p.n_mrows, feat, &P.hub_count_stride))) return rc;
            P.slot_hub = p.slot_hub.p; P.hub_count = c->hub_count.p;
        }
        P.probe = probe;
        P.unroll = 4;
        return launch_gcn_plan(P, c->stream);
    }
    if (mode == GNNAGG_MODE_BALANCED && c->partitions > 0 && c->plan_part.valid && c->part_descriptors) {
        // source-partitioned order on the plan kernel's short-row path: every group of sched[1] is a descriptor, rows with
        // several groups meet in scratch and k_combine folds them in ascending group order; tile-major on the tiled image
        // of X (2-D blocked) unless GNNAGG_TILED=0
        BalancedPlan &p = c->plan_part;
        TiledRun tr = plan_tiles(c, *s, x, y, feat, 4);
        const bool span_run = tr.spec.on && s->n_spans > 0;
        if (!span_run && s->gpu_built) {   // the descriptor form is needed after all: the host builder makes both
            c->force_host_plan = 1;
            if ((rc = build_partitioned(c, c->partitions))) return rc;
            return gcn_run(c, x, y, feat, mode, reduce, flags, nn, probe);
        }
        if (span_run) {  // every group owns a partial row (slot = group index): sequential flushes
            tr.spec.p_tile_stride = (long)s->num_target * tr.spec.tile_w;
            tr.partial_floats = (size_t)s->num_target * tr.spec.tile_w * tr.ntiles;
        }
        bool demoted = false;
        if ((rc = reserve_partitioned_scratch(c, tr.partial_floats, 0, tr.xt_floats, &demoted))) return rc;
        if (demoted) return gcn_run(c, x, y, feat, mode, reduce, flags, nn, probe);
        if ((rc = refresh_partitioned_val(c, s))) return rc;
        if (span_run) {
            SpanLaunch S;
            S.span_g = s->span_g.p; S.n_spans = s->n_spans; S.span_cost_prefix = s->span_cost_prefix.data();
            S.ptr_s = s->ptr_s.p; S.idx_f = s->idx_f.p; S.val_s = c->d_val ? s->val_s.p : nullptr; S.target = s->target.p;
            S.n_groups = s->num_target; S.crows = s->crows.p; S.n_crows = s->n_crows; S.rg_ptr = s->rg_ptr.p; S.rg_idx = s->rg_idx.p;
            S.empty_rows = s->empty_rows.p; S.n_empty = s->n_empty; S.row_ptr = c->d_ptr;
            S.x = x; S.x_rows = s->total_cols; S.y = y; S.partial = c->partial.p; S.feat = feat; S.reduce = reduce; S.relu = (flags & GNNAGG_FLAG_RELU) ? 1 : 0;
            S.tile = tr.spec; S.probe = probe;
            if (tr.retile) {
                if ((rc = launch_tile_x(x, c->xt.p, s->total_cols, feat, tr.spec.tile_w, c->stream))) return rc;
                S.x = c->xt.p;
            }
            if ((rc = launch_gcn_span(S, c->stream)) || !nn || probe) return rc;
            return launch_dense_nn(y, nn->weight, nn->out, c->V, nn->cols, feat, c->stream);
        }
        GcnPlanLaunch P;
        P.t0 = p.t0.p; P.n0 = p.n0; P.chunk = p.chunk; P.t0_cost_prefix = p.t0_cost_prefix.data();
        P.hubs = s->worklist();
        P.row_ptr = c->d_ptr; P.idx = s->idx_s.p; P.val = c->d_val ? s->val_s.p : nullptr; P.x = x; P.y = y; P.feat = feat; P.reduce = reduce;
        P.xcd_remap = c->xcd_remap; P.relu = (flags & GNNAGG_FLAG_RELU) ? 1 : 0; P.num_rows = c->V; P.t0_partials = 1;
        P.partial = c->partial.p;
        P.tile = tr.spec;
        if (tr.retile) {
            if ((rc = launch_tile_x(x, c->xt.p, s->total_cols, feat, tr.spec.tile_w, c->stream))) return rc;
            P.x = c->xt.p;
        }
        if (nn) { P.nn_weight = nn->weight; P.nn_out = nn->out; P.nn_cols = nn->cols; }
        P.probe = probe;
        return launch_gcn_plan(P, c->stream);
    }
    if (probe) return fail(GNNAGG_ERR_ARG, "probe: balanced mode (or a neighbor-grouping schedule that runs on the plan kernel) only");
    if (mode == GNNAGG_MODE_ROWS && c->use_plan) {
        if (!c->rows_plan.valid && (rc = build_rows_plan(c))) return rc;
        RowsPlan &p = c->rows_plan;
        const bool fork = p.n1 > 0 && c->use_aux_stream;
        if (p.n1 > 0) {  // long rows (disjoint output rows): forked to the auxiliary stream, or first on this one
            if (fork) {
                if (!c->aux_stream) {
                    HIP_TRY(hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking));
                    HIP_TRY(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
                    HIP_TRY(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
                }
                HIP_TRY(hipEventRecord(c->ev_fork, c->stream));
                HIP_TRY(hipStreamWaitEvent(c->aux_stream, c->ev_fork, 0));
            }
            GcnRowsLongLaunch R;
            R.tile_w = c->opt_hub_tile;
            R.r1 = p.r1.p; R.n1 = p.n1; R.idx = c->d_idx; R.val = c->d_val; R.x = x; R.y = y; R.feat = feat; R.reduce = reduce;
            R.relu = (flags & GNNAGG_FLAG_RELU) ? 1 : 0;
            if ((rc = launch_gcn_rows_long(R, fork ? c->aux_stream : c->stream))) return rc;
            if (fork) HIP_TRY(hipEventRecord(c->ev_join, c->aux_stream));
        }
        if (p.n2 > 0) {  // medium rows: ahead of the short rows on this stream (heaviest first; many workgroups per CU)
            GcnRowsLongLaunch R;
            R.tile_w = c->opt_hub_tile;
            R.r1 = p.r2.p; R.n1 = p.n2; R.idx = c->d_idx; R.val = c->d_val; R.x = x; R.y = y; R.feat = feat; R.reduce = reduce;
            R.relu = (flags & GNNAGG_FLAG_RELU) ? 1 : 0; R.medium = 1;
            if ((rc = launch_gcn_rows_long(R, c->stream))) return rc;
        }
        GcnPlanLaunch P;  // short rows: the descriptor path of the plan kernel (no segments, no hubs)
        P.t0 = p.r0.p; P.n0 = p.n0; P.t0_cost_prefix = p.r0_cost_prefix.data();
        P.row_ptr = c->d_ptr; P.idx = c->d_idx; P.val = c->d_val; P.x = x; P.y = y; P.feat = feat; P.reduce = reduce;
        P.xcd_remap = c->xcd_remap; P.num_rows = c->V; P.relu = (flags & GNNAGG_FLAG_RELU) ? 1 : 0;
        const bool nn_rows_ok = nn && (p.n1 + p.n2 == 0 || feat <= 15000);
        if (nn_rows_ok) {  // short rows: epilogue of the plan kernel (or the GEMM right behind it)
            P.nn_weight = nn->weight; P.nn_out = nn->out; P.nn_cols = nn->cols;
        }
        rc = launch_gcn_plan(P, c->stream);
        if (fork) HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_join, 0));  // join
        if (rc || !nn) return rc;
        if (!nn_rows_ok) return launch_dense_nn(y, nn->weight, nn->out, c->V, nn->cols, feat, c->stream);
        if (p.n1 + p.n2 > 0)  // the long and medium rows' products, once their chains have joined
            rc = launch_dense_rows(p.r1_rows.p, p.n1 + p.n2, y, nn->weight, nn->out, feat, nn->cols, c->stream);
        return rc;
    }
    GcnLaunch L;
    L.row_ptr = c->d_ptr; L.x = x; L.y = y; L.feat = feat; L.reduce = reduce;
    L.xcd_remap = c->xcd_remap; L.relu = (flags & GNNAGG_FLAG_RELU) ? 1 : 0;
    if (!s) {
        L.wl.ptr = c->d_ptr;
        L.wl.n_items = c->V;
        L.idx = c->d_idx;
        L.val = c->d_val;
        if (c->xcd_remap == 2) {
            if (c->row_cost_prefix.empty()) {
                if ((rc = fetch_host_ptr(c))) return rc;
                c->row_cost_prefix.resize((size_t)c->V + 1);
                for (int r = 0; r <= c->V; ++r) c->row_cost_prefix[r] = (long)c->h_ptr[r] + (long)kItemCost * r;
            }
            L.xcd_item_cost_prefix = c->row_cost_prefix.data();
        }
    } else {
        // permuted orders read a permuted copy of the edge values; it follows the caller's array (updateval re-aliases it,
        // aggr_gcn.h:540-544, and callers rewrite it in place), so it is re-gathered before every run
        if (s->permuted && (rc = refresh_partitioned_val(c, s))) return rc;
        L.xcd_item_cost_prefix = s->cost_prefix.data();
        L.wl = s->worklist();
        L.idx = s->permuted ? s->idx_s.p : c->d_idx;
        L.val = s->permuted ? (c->d_val ? s->val_s.p : nullptr) : c->d_val;
        if (s->n_slots > 0) {
            if ((rc = c->partial.reserve((size_t)s->n_slots * feat))) return rc;
            L.partial = c->partial.p;
        }
    }
    rc = launch_gcn(L, c->stream);
    if (rc || !nn) return rc;
    return launch_dense_nn(y, nn->weight, nn->out, c->V, nn->cols, feat, c->stream);
}

static int gat_run(Ctx *c, const float *x, const float *att, float *y, int feat, int heads, float slope, int mode,
                   float *newval, int probe = 0, int part = 0, float *den_io = nullptr)
{
    if (c->kind != Ctx::GAT) return fail(GNNAGG_ERR_ARG, "handle is not a GAT aggregator");
    if (!x || !y || !att) return fail(GNNAGG_ERR_ARG, "null feature/attention pointer");
    if (mode < GNNAGG_MODE_ROWS || mode > GNNAGG_MODE_BALANCED) return fail(GNNAGG_ERR_ARG, "bad mode");
    if (mode == GNNAGG_MODE_ROWS && c->fast_rows) mode = GNNAGG_MODE_BALANCED;
    if (mode == GNNAGG_MODE_SCHEDULED && c->fast_scheduled && c->sched[0].valid && !newval && sched_keeps_every_edge(c)) mode = GNNAGG_MODE_BALANCED;
    if (mode == GNNAGG_MODE_ROWS && c->opt_rows_blocked && c->tiled && c->use_plan && !newval && !probe && part == 0) {
        bool used = false;   // canonical chains on the blocked order where the graph allows it (sorted rows, high degree)
        const int rcb = run_rows_blocked_gat(c, x, att, y, feat, heads, slope, &used);
        if (rcb || used) return rcb;
    }
    Schedule *s = nullptr;
    int rc = get_sched(c, mode, &s);
    if (rc) return rc;
    if (part != 0) {  // two-pass form: the chunked plan kernel implements it (a handle on the 2-D blocked order gets a plan beside it)
        if (mode != GNNAGG_MODE_BALANCED || !c->use_plan || newval || probe || !den_io || part < 1 || part > 3)
            return fail(GNNAGG_ERR_ARG, "gat_run_part: GNNAGG_MODE_BALANCED, part 1, 2 or 3, a denominator array, no newval");
        if (!c->plan.valid && (rc = build_balanced_plan_keep(c))) return rc;
    }
    if (probe && !(mode == GNNAGG_MODE_BALANCED && c->partitions > 0 && c->plan_part.valid && c->part_descriptors))
        return fail(GNNAGG_ERR_ARG, "GAT probe: only the 2-D blocked balanced order has a probe instantiation");
    if ((mode == GNNAGG_MODE_BALANCED && c->use_plan && (c->partitions == 0 || part != 0)) || (mode == GNNAGG_MODE_SCHEDULED && c->plan_sched.valid)) {
        BalancedPlan &p = mode == GNNAGG_MODE_BALANCED ? c->plan : c->plan_sched;
        GatPlanLaunch P;
        P.t0 = p.t0.p; P.t1 = p.t1.p; P.n0 = p.n0; P.n1 = p.n1; P.chunk = p.chunk; P.t0_cost_prefix = p.t0_cost_prefix.data();
        if (wants_sorted_rows(c, feat)) {
            if ((rc = ensure_sorted_rows(c, p))) return rc;
            if (p.t0_sorted.p) { P.t0 = p.t0_sorted.p; P.t0_cost_prefix = p.t0s_cost_prefix.data(); }
        }
        P.hubs.mrow_id = p.mrow_id.p; P.hubs.mrow_ptr = p.mrow_ptr.p; P.hubs.n_mrows = p.n_mrows;
        P.hubs.n_slots = p.n_slots; P.hubs.big_rows = p.big_rows.p; P.hubs.n_big = p.n_big;
        P.idx = c->d_idx; P.att = att; P.x = x; P.y = y; P.newval = newval; P.feat = feat; P.heads = heads; P.slope = slope;
        P.xcd_remap = c->xcd_remap;
        if (p.n_slots > 0) {
            if ((rc = c->partial.reserve((size_t)p.n_slots * feat))) return rc;
            if ((rc = c->partial_den.reserve((size_t)p.n_slots * heads))) return rc;
            P.partial = c->partial.p;
            P.partial_den = c->partial_den.p;
        }
        if (p.n_mrows > 0 && c->inkernel_combine) {
            if ((rc = reserve_hub_counters(c, p.n_mrows, feat, &P.hub_count_stride))) return rc;
            P.slot_hub = p.slot_hub.p; P.hub_count = c->hub_count.p;
        }
        P.unroll = 4;
        P.part_mode = part; P.den_io = den_io;
        return launch_gat_plan(P, c->stream);
    }
    if (mode == GNNAGG_MODE_BALANCED && c->partitions > 0 && c->plan_part.valid && c->part_descriptors) {
        BalancedPlan &p = c->plan_part;  // source-partitioned order on the descriptor path, as in gcn_run
        if (heads <= 0 || feat % heads != 0) return fail(GNNAGG_ERR_ARG, "GAT needs feat % heads == 0");
        TiledRun tr = plan_tiles(c, *s, x, y, feat, feat / heads);
        const bool span_run = tr.spec.on && s->n_spans > 0 && gat_span_tiles(feat, heads, tr.spec.tile_w);
        if (!span_run && s->gpu_built) {   // head widths the span kernel does not tile: the descriptor form, from the host builder
            c->force_host_plan = 1;
            if ((rc = build_partitioned(c, c->partitions))) return rc;
            return gat_run(c, x, att, y, feat, heads, slope, mode, newval, probe, part, den_io);
        }
        size_t den_floats = (size_t)s->n_slots * heads;
        if (span_run) {
            tr.spec.p_tile_stride = (long)s->num_target * tr.spec.tile_w;
            tr.partial_floats = (size_t)s->num_target * tr.spec.tile_w * tr.ntiles;
            den_floats = (size_t)s->num_target * heads;
        }
        bool demoted = false;
        if ((rc = reserve_partitioned_scratch(c, tr.partial_floats, den_floats, tr.xt_floats, &demoted))) return rc;
        if (demoted) return gat_run(c, x, att, y, feat, heads, slope, mode, newval, probe);
        if (probe && !span_run) return fail(GNNAGG_ERR_ARG, "GAT probe: the segmented-stream kernel does not cover this width / head count");
        if (span_run) {
            GatSpanLaunch G;
            SpanLaunch &S = G.s;
            S.probe = probe;
            S.span_g = s->span_g.p; S.n_spans = s->n_spans; S.span_cost_prefix = s->span_cost_prefix.data();
            S.ptr_s = s->ptr_s.p; S.idx_f = s->idx_f.p; S.target = s->target.p;
            S.n_groups = s->num_target; S.crows = s->crows.p; S.n_crows = s->n_crows; S.rg_ptr = s->rg_ptr.p; S.rg_idx = s->rg_idx.p;
            S.empty_rows = s->empty_rows.p; S.n_empty = s->n_empty; S.row_ptr = c->d_ptr;
            S.x = x; S.x_rows = s->total_cols; S.y = y; S.partial = c->partial.p; S.feat = feat; S.tile = tr.spec;
            G.att = att; G.partial_den = c->partial_den.p; G.newval = newval; G.eperm = s->eperm.p; G.heads = heads; G.slope = slope;
            if (tr.retile) {
                if ((rc = launch_tile_x(x, c->xt.p, s->total_cols, feat, tr.spec.tile_w, c->stream))) return rc;
                S.x = c->xt.p;
            }
            {   // compact attention terms: one HT-float load per edge instead of HT strided ones (k_tile_att)
                const int dhead = feat / heads, ht = tr.spec.tile_w >= dhead ? tr.spec.tile_w / dhead : 1;
                const int n_hg = (heads + ht - 1) / ht, arows = c->V > s->total_cols ? c->V : s->total_cols;
                const size_t half = (size_t)n_hg * arows * ht;
                if ((rc = c->att_t.reserve(2 * half))) return rc;
                if ((rc = launch_tile_att(att, c->att_t.p, c->att_t.p + half, arows, heads, ht, c->stream))) return rc;
                G.as_t = c->att_t.p; G.ac_t = c->att_t.p + half; G.att_rows = arows;
            }
            return launch_gat_span(G, c->stream);
        }
        GatPlanLaunch P;
        P.t0 = p.t0.p; P.n0 = p.n0; P.chunk = p.chunk; P.t0_cost_prefix = p.t0_cost_prefix.data();
        P.hubs = s->worklist();
        P.idx = s->idx_s.p; P.att = att; P.x = x; P.y = y; P.newval = newval; P.feat = feat; P.heads = heads; P.slope = slope;
        P.eperm = s->eperm.p;  // newval[E,H] is defined in CSR edge order (gnnagg.h): scattered through the permutation
        P.xcd_remap = c->xcd_remap;
        P.partial = c->partial.p;
        P.partial_den = c->partial_den.p;
        P.tile = tr.spec;
        if (tr.retile) {
            if ((rc = launch_tile_x(x, c->xt.p, s->total_cols, feat, tr.spec.tile_w, c->stream))) return rc;
            P.x = c->xt.p;
        }
        return launch_gat_plan(P, c->stream);
    }
    if (mode == GNNAGG_MODE_ROWS && c->use_plan && !newval && heads > 0 && feat % heads == 0) {
        // canonical order: short rows on the descriptor path of k_gat_plan, isolated hub rows on the long-row kernel
        // (auxiliary stream) when a 32-column tile lies inside one head
        const bool tile_in_head = ((feat / heads) % 32) == 0;
        RowsPlan &p = tile_in_head ? c->rows_plan : c->rows_plan_nomed;
        if (!p.valid && (rc = build_rows_plan(c, tile_in_head))) return rc;
        const bool long_ok = p.n1 > 0 && tile_in_head;
        if (p.n1 + p.n2 == 0 || tile_in_head) {
            const bool fork = long_ok && c->use_aux_stream;
            if (long_ok) {
                if (fork) {
                    if (!c->aux_stream) {
                        HIP_TRY(hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking));
                        HIP_TRY(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
                        HIP_TRY(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
                    }
                    HIP_TRY(hipEventRecord(c->ev_fork, c->stream));
                    HIP_TRY(hipStreamWaitEvent(c->aux_stream, c->ev_fork, 0));
                }
                GcnRowsLongLaunch R;
            R.tile_w = c->opt_hub_tile;
                R.r1 = p.r1.p; R.n1 = p.n1; R.idx = c->d_idx; R.x = x; R.y = y; R.feat = feat;
                R.att = att; R.heads = heads; R.slope = slope;
                if ((rc = launch_gcn_rows_long(R, fork ? c->aux_stream : c->stream))) return rc;
                if (fork) HIP_TRY(hipEventRecord(c->ev_join, c->aux_stream));
            }
            if (p.n2 > 0) {  // medium rows, ahead of the short rows on this stream
                GcnRowsLongLaunch R;
            R.tile_w = c->opt_hub_tile;
                R.r1 = p.r2.p; R.n1 = p.n2; R.idx = c->d_idx; R.x = x; R.y = y; R.feat = feat;
                R.att = att; R.heads = heads; R.slope = slope; R.medium = 1;
                if ((rc = launch_gcn_rows_long(R, c->stream))) return rc;
            }
            GatPlanLaunch P;
            P.t0 = p.r0.p; P.n0 = p.n0; P.t0_cost_prefix = p.r0_cost_prefix.data();
            P.idx = c->d_idx; P.att = att; P.x = x; P.y = y; P.feat = feat; P.heads = heads; P.slope = slope;
            P.xcd_remap = c->xcd_remap; P.rows_semantics = 1;
            rc = launch_gat_plan(P, c->stream);
            if (fork) HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_join, 0));
            return rc;
        }
    }
    GatLaunch L;
    L.att = att; L.x = x; L.y = y; L.feat = feat; L.heads = heads; L.slope = slope; L.newval = newval;
    L.xcd_remap = c->xcd_remap;
    if (!s) {
        L.wl.ptr = c->d_ptr;
        L.wl.n_items = c->V;
        L.idx = c->d_idx;
    } else {
        L.wl = s->worklist();
        L.idx = s->permuted ? s->idx_s.p : c->d_idx;
        if (s->permuted && newval)
            return fail(GNNAGG_ERR_ARG, "newval output is defined for CSR edge order only (not locality schedules)");
        if (s->n_slots > 0) {
            if ((rc = c->partial.reserve((size_t)s->n_slots * feat))) return rc;
            if ((rc = c->partial_den.reserve((size_t)s->n_slots * heads))) return rc;
            L.partial = c->partial.p;
            L.partial_den = c->partial_den.p;
        }
    }
    return launch_gat(L, c->stream);
}

// Work items for the edge kernels: the balanced neighbor grouping (device arrays) of this handle.
static int edge_items(Ctx *c, Schedule **out)
{
    Schedule &s = c->sched_edges;
    if (!s.valid) {
        int rc = build_grouping(c, s, pick_chunk(c), GNNAGG_SCHED_NEIGHBOR_GROUPING);
        if (rc) return rc;
    }
    *out = &s;
    return GNNAGG_OK;
}

int edge_launch(Ctx *c, EdgeItemLaunch &L, int heads)
{
    Schedule *s = nullptr;
    int rc = edge_items(c, &s);
    if (rc) return rc;
    L.wl = s->worklist();
    L.idx = c->d_idx;
    L.heads = heads;
    L.avg_item_edges = s->num_target > 0 ? std::max(1, c->E / s->num_target) : 1;
    if (s->n_slots > 0) {
        if ((rc = c->partial_den.reserve((size_t)s->n_slots * heads))) return rc;
        L.partial_den = c->partial_den.p;
    }
    return GNNAGG_OK;
}

int do_schedule(Ctx *c, int kind, const int *param, int total_v)
{
    if (!param) return fail(GNNAGG_ERR_ARG, "null schedule parameter array");
    switch (kind) {
        case GNNAGG_SCHED_NEIGHBOR_GROUPING: {
            int rc = build_grouping(c, c->sched[0], param[0], kind);
            c->plan_sched.reset();
            if (rc == GNNAGG_OK && c->use_plan) {
                // the reference's groups of NG edges are the plan's chunks; use the plan kernel (in-workgroup ordered
                // fold) unless NG is so small that most rows would occupy a whole workgroup for a handful of edges
                double ratio = 0.0;
                rc = build_plan_into(c, c->plan_sched, param[0], false, &ratio);
                if (rc == GNNAGG_OK && ratio > 1.5) c->plan_sched.reset();
            }
            return rc;
        }
        case GNNAGG_SCHED_LOCALITY:
            c->plan_sched.reset();
            return build_locality(c, c->sched[0], param[0], 0, total_v, kind, true);
        case GNNAGG_SCHED_LOCALITY_NEIGHBOR_GROUPING:
            if (param[1] <= 0) return fail(GNNAGG_ERR_ARG, "neighbor group size must be >= 1");
            c->plan_sched.reset();
            return build_locality(c, c->sched[0], param[0], param[1], total_v, kind, true);
        case GNNAGG_SCHED_NOP:
            c->plan_sched.reset();
            c->sched[0].reset();
            return GNNAGG_OK;
        default:
            return fail(GNNAGG_ERR_ARG, "unknown schedule kind");
    }
}

void die_if_abort(int rc, const char *where)
{
    if (rc == GNNAGG_OK || !g_abort_on_error) return;
    // reference FatalError, include/util.h:82-92
    fprintf(stderr, "gnnagg failure in %s: %s\nAborting...\n", where, g_last_error.c_str());
    (void)hipDeviceReset();
    exit(1);
}

}  // namespace gnnagg

using namespace gnnagg;


#pragma GCC visibility push(default)
extern "C" {

const char *gnnagg_last_error(void) { return g_last_error.c_str(); }
int gnnagg_version(void) { return 100; }
void gnnagg_set_abort_on_error(int on) { g_abort_on_error = on != 0; }

// ------------------------------------------------------------------------------- Section B
static int create(Ctx::Kind kind, const int *d_ptr, const int *d_idx, const float *d_val, int V, int E,
                  gnnagg_handle *out)
{
    if (!out) return fail(GNNAGG_ERR_ARG, "null output handle");
    *out = 0;
    if (V < 0 || E < 0) return fail(GNNAGG_ERR_ARG, "negative graph size");
    if (!d_ptr || (E > 0 && !d_idx)) return fail(GNNAGG_ERR_ARG, "null CSR pointer");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(GNNAGG_ERR_HIP, "no HIP device available: libgnnagg has no CPU fallback");
    Ctx *c = new Ctx;
    c->kind = kind; c->V = V; c->E = E; c->d_ptr = d_ptr; c->d_idx = d_idx; c->d_val = d_val;
    if (const char *e = getenv("GNNAGG_XCD_REMAP")) c->xcd_remap = atoi(e);
#ifdef GNNAGG_EXTRAS
    if (const char *e = getenv("GNNAGG_PLAN")) c->use_plan = atoi(e);
#endif
    if (const char *e = getenv("GNNAGG_PARTITIONS")) c->opt_partitions = atoi(e);
    if (const char *e = getenv("GNNAGG_FAST_ROWS")) { c->fast_rows = atoi(e); c->fast_rows_from_env = true; }
    if (const char *e = getenv("GNNAGG_FAST_SCHEDULED")) c->fast_scheduled = atoi(e);
    if (const char *e = getenv("GNNAGG_AUX_STREAM")) c->use_aux_stream = atoi(e);
    {
        std::lock_guard<std::mutex> lk(g_mu);
        g_live.insert(c);
    }
    *out = reinterpret_cast<gnnagg_handle>(c);
    return GNNAGG_OK;
}

int gnnagg_gcn_create(const int *d_ptr, const int *d_idx, const float *d_val, int num_v, int num_e,
                      gnnagg_handle *out)
{
    return create(Ctx::GCN, d_ptr, d_idx, d_val, num_v, num_e, out);
}

int gnnagg_gat_create(const int *d_ptr, const int *d_idx, int num_v, int num_e, gnnagg_handle *out)
{
    return create(Ctx::GAT, d_ptr, d_idx, nullptr, num_v, num_e, out);
}

int gnnagg_destroy(gnnagg_handle h)
{
    Ctx *c;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        c = reinterpret_cast<Ctx *>(h);
        if (!g_live.count(c)) return fail(GNNAGG_ERR_ARG, "invalid or destroyed handle");
        g_live.erase(c);
    }
    (void)hipStreamSynchronize(c->stream);
#ifdef GNNAGG_EXTRAS
    if (c->tr.agg) (void)gnnagg_destroy(c->tr.agg);
#endif
    if (c->aux_stream) {
        (void)hipStreamSynchronize(c->aux_stream);
        (void)hipStreamDestroy(c->aux_stream);
        (void)hipEventDestroy(c->ev_fork);
        (void)hipEventDestroy(c->ev_join);
    }
    delete c;
    return GNNAGG_OK;
}

int gnnagg_set_stream(gnnagg_handle h, void *hip_stream)
{
    GET_CTX(h);
    c->stream = (hipStream_t)hip_stream;
    return GNNAGG_OK;
}

int gnnagg_set_option(gnnagg_handle h, const char *name, int value)
{
    GET_CTX(h);
    if (!name) return fail(GNNAGG_ERR_ARG, "null option name");
    const std::string n(name);
    bool replan = false;  // the library-chosen balanced order depends on it: rebuilt on the next use
    if (n == "partitions") { if (value < -1) return fail(GNNAGG_ERR_ARG, "partitions: -1 (auto), 0 (never) or a count"); c->opt_partitions = value; c->no_auto_partition = 0; replan = true; }
#ifdef GNNAGG_EXTRAS
    else if (n == "partition_min_degree") { c->opt_part_min_deg = value; replan = true; }
    else if (n == "retile") c->opt_retile = value;
    else if (n == "tiled") c->tiled = value;
    else if (n == "spans") { c->use_spans = value; replan = true; }
    else if (n == "inkernel_combine") c->inkernel_combine = value;
    else if (n == "host_plan") replan = true;
#endif
    else if (n == "tile_width") { if (value != 32 && value != 64 && value != 128 && value != 256) return fail(GNNAGG_ERR_ARG, "tile_width: 32, 64, 128 or 256"); c->opt_tile_w = value; replan = true; }
    else if (n == "slice_kb") { if (value < 1) return fail(GNNAGG_ERR_ARG, "slice_kb must be >= 1"); c->opt_slice_kb = value; replan = true; }
    else if (n == "scratch_limit_mb") c->opt_scratch_limit_mb = value;
    else if (n == "fast_rows") c->fast_rows = value;
    else if (n == "reference_defaults") {
        // what a handle made through a reference-facing surface starts with: run(vin, vout, B, 0) takes the balanced order
        // (the environment still has the last word)
        if (!c->fast_rows_from_env) c->fast_rows = value != 0;
    }
    else if (n == "fast_scheduled") c->fast_scheduled = value;
    else if (n == "aux_stream") c->use_aux_stream = value;
    else if (n == "rows_blocked") c->opt_rows_blocked = value;
    else if (n == "rows_hub_edges") { c->opt_rb_hub_edges = std::max(0, value); replan = true; }
    else if (n == "rows_hub_tile") {
        if (value != 0 && value != 32 && value != 64) return fail(GNNAGG_ERR_ARG, "rows_hub_tile: 0, 32 or 64");
        c->opt_hub_tile = value;
    }
    else if (n == "rows_medium_edges") { c->opt_rows_medium = value; c->rows_plan.valid = false; c->rows_plan_nomed.valid = false; }
    else return fail(GNNAGG_ERR_ARG, "unknown option: " + n);
    if (replan) {
        c->force_host_plan = (n == "host_plan") ? (value != 0) : 0;
        c->partitions = 0;
        c->sched[1].reset();
        c->plan_part.reset();
        c->plan.reset();
        c->rb.reset();
    }
    return GNNAGG_OK;
}

int gnnagg_update_val(gnnagg_handle h, const float *d_val)
{
    GET_CTX(h);
    if (c->kind != Ctx::GCN) return fail(GNNAGG_ERR_ARG, "handle is not a GCN aggregator");
    c->d_val = d_val;  // aliases, like aggr_gcn.h:540-544; permuted schedules re-gather their copy before every run
    return GNNAGG_OK;
}

int gnnagg_set_row_aux(gnnagg_handle h, const int *d_row_aux)
{
    GET_CTX(h);
    if (c->kind != Ctx::GCN) return fail(GNNAGG_ERR_ARG, "handle is not a GCN aggregator");
    c->row_aux = d_row_aux;  // borrowed, read at run time
    return GNNAGG_OK;
}

int gnnagg_schedule(gnnagg_handle h, int kind, const int *param, int total_num_v)
{
    GET_CTX(h);
    return do_schedule(c, kind, param, total_num_v);
}

int gnnagg_schedule_balanced(gnnagg_handle h, int chunk)
{
    GET_CTX(h);
    if (chunk < 0) return fail(GNNAGG_ERR_ARG, "chunk must be >= 0");
    c->partitions = 0;
    c->sched[1].reset();
    c->plan_part.reset();
    if (chunk == 0 && c->use_plan && auto_partitions(c) != 0) return build_partitioned(c, auto_partitions(c));
    if (c->use_plan) return build_balanced_plan(c, chunk > 0 ? chunk : pick_chunk(c));
    return build_grouping(c, c->sched[1], chunk > 0 ? chunk : pick_chunk(c), GNNAGG_SCHED_NEIGHBOR_GROUPING);
}

int gnnagg_mode_params(gnnagg_handle h, int mode, int *chunk, int *seg_chunks)
{
    GET_CTX(h);
    if (mode == GNNAGG_MODE_ROWS && c->fast_rows) mode = GNNAGG_MODE_BALANCED;
    if (mode == GNNAGG_MODE_ROWS) {
        if (chunk) *chunk = 0x7fffffff;
        if (seg_chunks) *seg_chunks = 0;
        return GNNAGG_OK;
    }
    Schedule *s = nullptr;
    int rc = get_sched(c, mode, &s);
    if (rc) return rc;
    const bool plan = mode == GNNAGG_MODE_BALANCED ? (c->use_plan != 0 && c->partitions == 0) : c->plan_sched.valid;
    if (chunk) {
        int mx = 0;
        if (plan) mx = mode == GNNAGG_MODE_BALANCED ? c->plan.chunk : c->plan_sched.chunk;
        else for (int g = 0; g < s->num_target; ++g) mx = std::max(mx, s->h_ptr_s[g + 1] - s->h_ptr_s[g]);
        *chunk = mx;
    }
    if (seg_chunks) *seg_chunks = plan ? kSegChunksHost : 0;
    return GNNAGG_OK;
}

int gnnagg_balanced_params(gnnagg_handle h, int *chunk, int *seg_chunks)
{
    return gnnagg_mode_params(h, GNNAGG_MODE_BALANCED, chunk, seg_chunks);
}

int gnnagg_balanced_partitions(gnnagg_handle h, int *partitions, int *total_cols)
{
    GET_CTX(h);
    if (!partitions) return fail(GNNAGG_ERR_ARG, "null output");
    Schedule *s = nullptr;
    int rc = get_sched(c, GNNAGG_MODE_BALANCED, &s);
    if (rc) return rc;
    *partitions = c->partitions;
    if (total_cols) *total_cols = c->partitions > 0 ? c->sched[1].total_cols : 0;
    return GNNAGG_OK;
}

int gnnagg_plan_info(gnnagg_handle h, double *plan_seconds, double *rows_plan_seconds, long long *plan_bytes, long long *scratch_bytes)
{
    GET_CTX(h);
    if (plan_seconds) *plan_seconds = c->plan_seconds;
    if (rows_plan_seconds) *rows_plan_seconds = c->rb_plan_seconds;
    if (plan_bytes)
        *plan_bytes = (long long)(c->plan_bytes + sched_device_bytes(c->rb.sched) + (c->rb.span_g.n + c->rb.idx_f.n + c->rb.r1.n) * sizeof(int) + c->rb.hub_mask.n);
    if (scratch_bytes) *scratch_bytes = (long long)((c->partial.n + c->partial_den.n + c->xt.n + c->yt.n + c->den_t.n + c->att_t.n + c->den.n) * sizeof(float));
    return GNNAGG_OK;
}

int gnnagg_rows_blocked_ranges(gnnagg_handle h, int *ranges)
{
    GET_CTX(h);
    if (!ranges) return fail(GNNAGG_ERR_ARG, "null output");
    *ranges = 0;
    if (!c->opt_rows_blocked || !c->tiled || !c->use_plan || c->fast_rows) return GNNAGG_OK;
    int rc;
    if (!c->rb.tried && (rc = build_rows_blocked(c, 4))) return rc;
    if (c->rb.ok) *ranges = c->rb.sched.par_num;
    return GNNAGG_OK;
}

int gnnagg_num_target(gnnagg_handle h, int mode, int *out)
{
    GET_CTX(h);
    if (!out) return fail(GNNAGG_ERR_ARG, "null output");
    if (mode == GNNAGG_MODE_ROWS && c->fast_rows) mode = GNNAGG_MODE_BALANCED;
    if (mode == GNNAGG_MODE_ROWS) {
        *out = c->V;
        return GNNAGG_OK;
    }
    Schedule *s = nullptr;
    int rc = get_sched(c, mode, &s);
    if (rc) return rc;
    *out = s->num_target;
    return GNNAGG_OK;
}

int gnnagg_get_schedule(gnnagg_handle h, int mode, int *h_ptr_s, int *h_idx_s, int *h_target, float *h_val_s)
{
    GET_CTX(h);
    Schedule *s = nullptr;
    int rc = get_sched(c, mode, &s);
    if (rc) return rc;
    if (!s) return fail(GNNAGG_ERR_ARG, "MODE_ROWS has no schedule");
    const int G = s->num_target;
    if (h_ptr_s) memcpy(h_ptr_s, s->h_ptr_s.data(), ((size_t)G + 1) * sizeof(int));
    if (h_target && G > 0) memcpy(h_target, s->h_target.data(), (size_t)G * sizeof(int));
    const size_t ne = s->gpu_built ? (size_t)s->n_edges_perm : s->permuted ? s->h_idx_s.size() : (size_t)c->E;
    if (h_idx_s && ne > 0) {
        if (int rc_ = copy_to_host(c, h_idx_s, s->gpu_built ? s->idx_f.p : s->permuted ? s->idx_s.p : c->d_idx, ne * sizeof(int))) return rc_;
        if (s->gpu_built)   // the device holds the ids with the span kernel's two flag bits
            for (size_t e = 0; e < ne; ++e) h_idx_s[e] &= 0x3fffffff;
    }
    if (h_val_s && ne > 0) {
        if (s->permuted && s->eperm.p && c->d_val) {   // library-built orders gather their copy of the values before every run
            if ((rc = refresh_partitioned_val(c, s))) return rc;
            HIP_TRY(hipStreamSynchronize(c->stream));
        }
        const float *src = s->permuted ? s->val_s.p : c->d_val;
        if (!src) return fail(GNNAGG_ERR_STATE, "aggregator has no edge values");
        if (int rc_ = copy_to_host(c, h_val_s, src, ne * sizeof(float))) return rc_;
    }
    return GNNAGG_OK;
}

int gnnagg_gcn_run(gnnagg_handle h, const float *d_x, float *d_y, int feat, int mode, int reduce)
{
    GET_CTX(h);
    return gcn_run(c, d_x, d_y, feat, mode, reduce);
}

int gnnagg_matmul_nn(const float *d_a, const float *d_b, float *d_c, int m, int n, int k, void *hip_stream)
{
    if (m < 0 || n < 0 || k < 0 || ((long)m * n > 0 && !d_c) || ((long)m * k > 0 && !d_a) || ((long)k * n > 0 && !d_b))
        return fail(GNNAGG_ERR_ARG, "bad matmul_nn arguments");
    return launch_dense_nn(d_a, d_b, d_c, m, n, k, hip_stream);
}

int gnnagg_gcn_run_with_nn(gnnagg_handle h, const float *d_x, float *d_y, const float *d_weight, float *d_transformed,
                           int feat_in, int feat_out, int mode)
{
    GET_CTX(h);
    if (!d_weight || !d_transformed || feat_out <= 0) return fail(GNNAGG_ERR_ARG, "bad run_with_nn arguments");
    const NnRequest nn = {d_weight, d_transformed, feat_out};
    return gcn_run(c, d_x, d_y, feat_in, mode, GNNAGG_REDUCE_SUM, 0, &nn);
}

int gnnagg_gcn_run_clock(gnnagg_handle h, const float *d_x, float *d_y, int feat, int mode, unsigned long long *d_timer,
                         int *num_blocks, int *waves_per_cu)
{
    GET_CTX(h);
    if (c->kind != Ctx::GCN) return fail(GNNAGG_ERR_ARG, "handle is not a GCN aggregator");
    if (mode != GNNAGG_MODE_ROWS && mode != GNNAGG_MODE_SCHEDULED) return fail(GNNAGG_ERR_ARG, "run_clock: mode must be rows or scheduled");
    if (!num_blocks) return fail(GNNAGG_ERR_ARG, "null num_blocks");
    if (d_timer && (!d_x || !d_y)) return fail(GNNAGG_ERR_ARG, "null feature pointer");
    // the instrumented kernel runs one work item per lane group: the CSR rows themselves (rows) or the user's groups
    // (scheduled) -- "fast_rows" / "fast_scheduled" do not apply here, the load-balance study is about exactly these two
    Schedule *s = nullptr;
    int rc = GNNAGG_OK;
    if (mode == GNNAGG_MODE_SCHEDULED && (rc = get_sched(c, mode, &s))) return rc;
    GcnLaunch L;
    L.row_ptr = c->d_ptr; L.x = d_x; L.y = d_y; L.feat = feat; L.reduce = GNNAGG_REDUCE_SUM; L.xcd_remap = 0;
    L.timer = d_timer; L.timer_blocks_out = num_blocks;
    L.timer_capacity = d_timer ? *num_blocks : 0;  // in: workgroups d_timer has room for (the size query's answer)
    if (!s) {
        L.wl.ptr = c->d_ptr; L.wl.n_items = c->V; L.idx = c->d_idx; L.val = c->d_val;
    } else {
        L.wl = s->worklist();
        L.idx = s->permuted ? s->idx_s.p : c->d_idx;
        L.val = s->permuted ? s->val_s.p : c->d_val;
        if (s->n_slots > 0 && d_timer) {
            if ((rc = c->partial.reserve((size_t)s->n_slots * feat))) return rc;
            L.partial = c->partial.p;
        }
    }
    if (waves_per_cu) *waves_per_cu = 32;  // hardware limit; the items kernel's register budget allows 24-28
    return launch_gcn(L, c->stream);
}

long long gnnagg_wall_clock_hz(void)
{
    int khz = 0, devid = 0;
    if (hipGetDevice(&devid) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, devid) != hipSuccess)
        return 100000000LL;
    return (long long)khz * 1000LL;
}

int gnnagg_gcn_run_ex(gnnagg_handle h, const float *d_x, float *d_y, int feat, int mode, int reduce, int flags)
{
    GET_CTX(h);
    return gcn_run(c, d_x, d_y, feat, mode, reduce, flags);
}

int gnnagg_gcn_probe_gather(gnnagg_handle h, const float *d_x, int feat, int mode)
{
    GET_CTX(h);
    // y is never written by the probe instantiation; a non-null pointer keeps the argument checks and the lane geometry
    // (alignment class of y) those of a real run
    return gcn_run(c, d_x, const_cast<float *>(d_x), feat, mode, GNNAGG_REDUCE_SUM, 0, nullptr, 1);
}

int gnnagg_check_csr(gnnagg_handle h, int num_cols, int *bad_rows, int *bad_indices)
{
    GET_CTX(h);
    int *d_counts = nullptr, counts[2] = {0, 0};
    HIP_TRY(hipMalloc((void **)&d_counts, 2 * sizeof(int)));
    int rc = launch_check_csr(c->d_ptr, c->d_idx, c->V, c->E, num_cols > 0 ? num_cols : c->V, d_counts, c->stream);
    if (!rc) {
        hipError_t e = hipMemcpyAsync(counts, d_counts, sizeof(counts), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) rc = fail(GNNAGG_ERR_HIP, hipGetErrorString(e));
    }
    (void)hipFree(d_counts);
    if (bad_rows) *bad_rows = counts[0];
    if (bad_indices) *bad_indices = counts[1];
    return rc;
}

int gnnagg_csr2edgelist(gnnagg_handle h, int *d_edgelist)
{
    GET_CTX(h);
    if (!d_edgelist && c->E > 0) return fail(GNNAGG_ERR_ARG, "null edge list");
    return launch_csr2edgelist(c->d_ptr, c->d_idx, d_edgelist, c->V, c->avg_deg(), c->stream);
}

int gnnagg_gcn_run_edgewise(gnnagg_handle h, const float *d_x, float *d_y, int feat)
{
    GET_CTX(h);
    if (c->kind != Ctx::GCN) return fail(GNNAGG_ERR_ARG, "handle is not a GCN aggregator");
    if (!d_x || !d_y || feat <= 0) return fail(GNNAGG_ERR_ARG, "bad edge-wise arguments");
    int rc;
    if (c->edgelist.n < (size_t)2 * c->E || c->edgelist.p == nullptr) {
        if ((rc = c->edgelist.reserve((size_t)2 * std::max(c->E, 1)))) return rc;
        if ((rc = launch_csr2edgelist(c->d_ptr, c->d_idx, c->edgelist.p, c->V, c->avg_deg(), c->stream))) return rc;
    }
    return launch_edgewise(c->edgelist.p, c->d_val, d_x, d_y, c->E, c->V, feat, c->stream);
}

int gnnagg_gat_run(gnnagg_handle h, const float *d_x, const float *d_att, float *d_y, int feat, int heads,
                   float slope, int mode, float *d_newval)
{
    GET_CTX(h);
    return gat_run(c, d_x, d_att, d_y, feat, heads, slope, mode, d_newval);
}

int gnnagg_gat_run_part(gnnagg_handle h, const float *d_x, const float *d_att, float *d_y, int feat, int heads, float slope, int part,
                        float *d_den_io)
{
    GET_CTX(h);
    return gat_run(c, d_x, d_att, d_y, feat, heads, slope, GNNAGG_MODE_BALANCED, nullptr, 0, part, d_den_io);
}

int gnnagg_gat_probe_gather(gnnagg_handle h, const float *d_x, const float *d_att, int feat, int heads, int mode)
{
    GET_CTX(h);
    // y is never written by the probe instantiation (see gnnagg_gcn_probe_gather)
    return gat_run(c, d_x, d_att, const_cast<float *>(d_x), feat, heads, 0.2f, mode, nullptr, 1);
}

int gnnagg_probe_row_gather(const void *d_rows, long long pitch_bytes, int seg_bytes, const int *d_ids, long long n_ids, int ids_per_group,
                            void *hip_stream)
{
    return launch_probe_row_gather(d_rows, (long)pitch_bytes, seg_bytes, d_ids, (long)n_ids, ids_per_group, hip_stream);
}

int gnnagg_gat_run_att(gnnagg_handle h, const float *d_att, float *d_out_val, int heads, float slope)
{
    GET_CTX(h);
    if (!d_att || (!d_out_val && c->E > 0) || heads <= 0) return fail(GNNAGG_ERR_ARG, "bad run_att arguments");
    EdgeItemLaunch L;
    int rc = edge_launch(c, L, heads);
    if (rc) return rc;
    if ((rc = c->den.reserve((size_t)std::max(c->V, 1) * heads))) return rc;
    L.att = d_att; L.out = d_out_val; L.den = c->den.p; L.slope = slope;
    if ((rc = launch_edge_items_sum(L, 0, c->stream))) return rc;  // w_e and row sums   (attGat :13-25)
    L.in = c->den.p;
    return launch_edge_items_map(L, 0, c->stream);                 // w_e / sum          (attGat :26-29)
}

int gnnagg_gat_run_u_add_v(gnnagg_handle h, const float *d_att, float *d_out_val)
{
    GET_CTX(h);
    if (!d_att || (!d_out_val && c->E > 0)) return fail(GNNAGG_ERR_ARG, "bad u_add_v arguments");
    EdgeItemLaunch L;
    int rc = edge_launch(c, L, 1);
    if (rc) return rc;
    L.att = d_att; L.out = d_out_val;
    return launch_edge_items_map(L, 1, c->stream);
}

int gnnagg_gat_run_add_to_center(gnnagg_handle h, const float *d_in_val, float *d_out_att)
{
    GET_CTX(h);
    if ((!d_in_val && c->E > 0) || !d_out_att) return fail(GNNAGG_ERR_ARG, "bad add_to_center arguments");
    EdgeItemLaunch L;
    int rc = edge_launch(c, L, 1);
    if (rc) return rc;
    L.in = d_in_val; L.den = d_out_att;
    return launch_edge_items_sum(L, 1, c->stream);
}

int gnnagg_gat_run_div_each(gnnagg_handle h, const float *d_in_att, float *d_inout_val)
{
    GET_CTX(h);
    if (!d_in_att || (!d_inout_val && c->E > 0)) return fail(GNNAGG_ERR_ARG, "bad div_each arguments");
    EdgeItemLaunch L;
    int rc = edge_launch(c, L, 1);
    if (rc) return rc;
    L.in = d_in_att; L.out = d_inout_val;
    return launch_edge_items_map(L, 0, c->stream);
}

int gnnagg_spmm_naive(const int *d_ptr, const int *d_idx, const float *d_val, const float *d_x, float *d_y,
                      int num_v, int feat, void *hip_stream)
{
    if (!d_ptr || !d_val || !d_x || !d_y || feat <= 0 || num_v < 0) return fail(GNNAGG_ERR_ARG, "bad spmm arguments");
    return launch_spmm_naive(d_ptr, d_idx, d_val, d_x, d_y, num_v, feat, hip_stream);
}

static int count_result(int *d_diff, int *h_diff, void *stream)
{
    HIP_TRY(hipMemcpyAsync(h_diff, d_diff, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    (void)hipFree(d_diff);
    return GNNAGG_OK;
}

int gnnagg_validate(const float *d_ref, const float *d_ans, int num, int *h_diff, void *hip_stream)
{
    if (!d_ref || !d_ans || !h_diff || num < 0) return fail(GNNAGG_ERR_ARG, "bad validate arguments");
    int *d_diff = nullptr;
    HIP_TRY(hipMalloc((void **)&d_diff, sizeof(int)));
    int rc = launch_validate(d_ref, d_ans, num, d_diff, hip_stream);
    if (rc) { (void)hipFree(d_diff); return rc; }
    return count_result(d_diff, h_diff, hip_stream);
}

int gnnagg_validate_reordered(const float *d_ref, const float *d_ans, const int *d_map, int num_v, int feat,
                              int *h_diff, void *hip_stream)
{
    if (!d_ref || !d_ans || !d_map || !h_diff || num_v < 0 || feat <= 0)
        return fail(GNNAGG_ERR_ARG, "bad validate_reordered arguments");
    int *d_diff = nullptr;
    HIP_TRY(hipMalloc((void **)&d_diff, sizeof(int)));
    int rc = launch_validate_reordered(d_ref, d_ans, d_map, num_v, feat, d_diff, hip_stream);
    if (rc) { (void)hipFree(d_diff); return rc; }
    return count_result(d_diff, h_diff, hip_stream);
}

}  // extern "C"
#pragma GCC visibility pop
