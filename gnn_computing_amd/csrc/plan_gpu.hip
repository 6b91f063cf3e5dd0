// plan_gpu.hip -- the 2-D blocked order's plans built ON THE DEVICE.
//
// The order is the reference's: localityNeighborGrouping (graph_schedule.h:156-243) for the balanced mode -- per source range, per
// row, the sub-row of edges whose source falls in the range, cut every NG edges -- and locality_schedule (:17-63, no cut) for the
// chained rows mode.  Round 2 / 3 built it on the host (host_graph.cpp) from a D2H copy of the CSR and uploaded 2-3 GB of permuted
// arrays: 1.1 s (balanced) / 2.8 s (chains) for the reddit-shaped graph, 70-180 aggregation steps.  Here the per-edge work never
// leaves HBM:
//   range key per edge -> STABLE radix sort of the edge ids by range (the CSR order inside a range IS row-major, in-row order kept:
//   exactly the reference's order) -> sub-row starts by comparing neighbours -> max-scan (start of the enclosing sub-row) -> group
//   starts every NG positions -> prefix sum = group ids -> scatter ptr_s / target -> row -> groups lists by a stable sort of the
//   group ids by row -> flagged ids.  The sort and the scans are prims.cuh (hand-written, wave64).
// Only per-GROUP arrays (a few MB) come to the host: the span cut is a greedy walk and the schedule queries read them.
// Results are identical to the host builders' (tests/test_gpu_blocked.py compares both with the oracle's restatement).
#include <algorithm>

#include "devbuf.h"
#include "prims.cuh"

namespace gnnagg {

namespace {

constexpr int kThreads = 256;
inline unsigned blocks_for(long n) { return (unsigned)((n + kThreads - 1) / kThreads); }

__global__ void k_range_keys(const int *__restrict__ idx, int E, int width, int P, unsigned *__restrict__ key, int *__restrict__ val)
{
    const int e = blockIdx.x * kThreads + threadIdx.x;
    if (e >= E) return;
    const int p = idx[e] / width;
    key[e] = (unsigned)(p >= P ? P - 1 : p);
    val[e] = e;
}

// mark[ptr[r]] = r for every non-empty row (mark zero-filled): an inclusive max-scan turns it into the row of every edge
__global__ void k_mark_row_starts(const int *__restrict__ ptr, int V, int *__restrict__ mark)
{
    const int r = blockIdx.x * kThreads + threadIdx.x;
    if (r < V && ptr[r + 1] > ptr[r]) mark[ptr[r]] = r;
}

__global__ void k_rows_unsorted(const int *__restrict__ idx, const int *__restrict__ row_of, int E, int *__restrict__ flag)
{
    const int e = blockIdx.x * kThreads + threadIdx.x;
    if (e <= 0 || e >= E) return;
    if (row_of[e] == row_of[e - 1] && idx[e] < idx[e - 1]) *flag = 1;
}

__global__ void k_gather_sorted(const int *__restrict__ eperm, const int *__restrict__ row_of, const int *__restrict__ idx, int E,
                                int *__restrict__ row_s, int *__restrict__ idx_s)
{
    const int pos = blockIdx.x * kThreads + threadIdx.x;
    if (pos >= E) return;
    const int e = eperm[pos];
    row_s[pos] = row_of[e];
    idx_s[pos] = idx[e];
}

// ss[pos] = pos where a (row, range) sub-row starts, else 0: an inclusive max-scan gives every position the start of its sub-row
__global__ void k_subrow_starts(const unsigned *__restrict__ key_s, const int *__restrict__ row_s, int E, int *__restrict__ ss)
{
    const int pos = blockIdx.x * kThreads + threadIdx.x;
    if (pos >= E) return;
    const bool start = pos == 0 || key_s[pos] != key_s[pos - 1] || row_s[pos] != row_s[pos - 1];
    ss[pos] = start ? pos : 0;
}

__global__ void k_group_flags(const int *__restrict__ ss, int E, int ng, int *__restrict__ gflag)
{
    const int pos = blockIdx.x * kThreads + threadIdx.x;
    if (pos >= E) return;
    gflag[pos] = ng > 0 ? ((pos - ss[pos]) % ng == 0) : (ss[pos] == pos);
}

__global__ void k_scatter_groups(const int *__restrict__ gflag, const int *__restrict__ gid, const int *__restrict__ row_s,
                                 const unsigned *__restrict__ key_s, int E, int G, int *__restrict__ ptr_s, int *__restrict__ target,
                                 unsigned *__restrict__ gkey)
{
    const int pos = blockIdx.x * kThreads + threadIdx.x;
    if (pos == 0) ptr_s[G] = E;
    if (pos >= E || !gflag[pos]) return;
    const int g = gid[pos];
    ptr_s[g] = pos;
    target[g] = row_s[pos];
    if (gkey) gkey[g] = key_s[pos];
}

__global__ void k_count_groups(const int *__restrict__ target, int G, int *__restrict__ groups_of)
{
    const int g = blockIdx.x * kThreads + threadIdx.x;
    if (g < G) atomicAdd(&groups_of[target[g]], 1);
}

__global__ void k_iota(int *__restrict__ a, int n)
{
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i < n) a[i] = i;
}

// ids with the group-end flag in bit 31 and the single-group-row flag in bit 30 (agg_span.hip)
__global__ void k_flag_ids(const int *__restrict__ idx_s, const int *__restrict__ gflag, const int *__restrict__ row_s,
                           const int *__restrict__ groups_of, int E, int *__restrict__ idx_f)
{
    const int pos = blockIdx.x * kThreads + threadIdx.x;
    if (pos >= E) return;
    unsigned w = (unsigned)idx_s[pos];
    if (pos == E - 1 || gflag[pos + 1]) {
        w |= 0x80000000u;
        if (groups_of[row_s[pos]] == 1) w |= 0x40000000u;
    }
    idx_f[pos] = (int)w;
}

// ---- chain plan
__global__ void k_mark_hub_rows(const int *__restrict__ ptr0, const int *__restrict__ target0, int G0, int hub_edges, unsigned char *__restrict__ is_hub)
{
    const int g = blockIdx.x * kThreads + threadIdx.x;
    if (g < G0 && ptr0[g + 1] - ptr0[g] > hub_edges) is_hub[target0[g]] = 1;   // (several groups may store the same 1)
}

// sort keys of a group (two stable 32-bit passes: length descending, then range): range major, longest sub-row first inside a range;
// groups of hub rows go behind everything
__global__ void k_chain_keys(const int *__restrict__ ptr0, const int *__restrict__ target0, const unsigned *__restrict__ gkey0, int G0, int P,
                             const unsigned char *__restrict__ is_hub, unsigned *__restrict__ key_len, unsigned *__restrict__ key_range, int *__restrict__ keep)
{
    const int g = blockIdx.x * kThreads + threadIdx.x;
    if (g >= G0) return;
    const bool hub = is_hub[target0[g]] != 0;
    key_len[g] = 0xffffffffu - (unsigned)(ptr0[g + 1] - ptr0[g]);
    key_range[g] = hub ? (unsigned)P : gkey0[g];
    keep[g] = hub ? 0 : 1;
}

__global__ void k_gather_u32(const unsigned *__restrict__ src, const int *__restrict__ idx, int n, unsigned *__restrict__ dst)
{
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}

// new order k -> old group ord[k]; lens[k] = its edges (0 for the dropped tail), inv[old g] = k
__global__ void k_chain_lens(const int *__restrict__ ord, const int *__restrict__ ptr0, int G0, int G, int *__restrict__ lens, int *__restrict__ inv)
{
    const int k = blockIdx.x * kThreads + threadIdx.x;
    if (k >= G0) return;
    const int g = ord[k];
    inv[g] = k < G ? k : -1;
    lens[k] = k < G ? ptr0[g + 1] - ptr0[g] : 0;
}

__global__ void k_chain_groups(const int *__restrict__ ord, const int *__restrict__ target0, const unsigned *__restrict__ gkey0, int G,
                               int *__restrict__ target, unsigned *__restrict__ grange)
{
    const int k = blockIdx.x * kThreads + threadIdx.x;
    if (k >= G) return;
    target[k] = target0[ord[k]];
    grange[k] = gkey0[ord[k]];
}

__global__ void k_chain_edges(const int *__restrict__ gflag, const int *__restrict__ gid, const int *__restrict__ inv, const int *__restrict__ ptr0,
                              const int *__restrict__ nptr, const int *__restrict__ idx_s, const int *__restrict__ eperm0, int E,
                              int *__restrict__ idx_f, int *__restrict__ eperm)
{
    const int pos = blockIdx.x * kThreads + threadIdx.x;
    if (pos >= E) return;
    const int g = gid[pos] + gflag[pos] - 1;     // inclusive count - 1 = the group of this position
    const int k = inv[g];
    if (k < 0) return;
    const int np = nptr[k] + (pos - ptr0[g]);
    unsigned w = (unsigned)idx_s[pos];
    if (pos == E - 1 || gflag[pos + 1]) w |= 0x80000000u;
    idx_f[np] = (int)w;
    eperm[np] = eperm0[pos];
}

struct Sorted {   // the edges in (range, CSR) order
    DevBuf<int> eperm, row_s, idx_s, row_of;
    DevBuf<unsigned> key_s;
};

struct Tmp { DevBuf<int> counts, scan; };   // scratch of the primitives (grow-only across the calls of one build)

int sort_edges_by_range(const int *d_ptr, const int *d_idx, int V, int E, int P, int width, hipStream_t st, Sorted &o, Tmp &tmp)
{
    int rc;
    DevBuf<unsigned> key;
    DevBuf<int> val;
    if ((rc = key.alloc((size_t)E)) || (rc = val.alloc((size_t)E)) || (rc = o.key_s.alloc((size_t)E)) || (rc = o.eperm.alloc((size_t)E)) ||
        (rc = o.row_of.alloc((size_t)E)) || (rc = o.row_s.alloc((size_t)E)) || (rc = o.idx_s.alloc((size_t)E)))
        return rc;
    hipLaunchKernelGGL(k_range_keys, dim3(blocks_for(E)), dim3(kThreads), 0, st, d_idx, E, width, P, key.p, val.p);
    int bits = 1;
    while ((1 << bits) < P) ++bits;
    if ((rc = prims::sort_pairs(key.p, o.key_s.p, val.p, o.eperm.p, E, bits, tmp.counts, tmp.scan, st))) return rc;   // stable: CSR order inside a range
    HIP_TRY(hipMemsetAsync(o.row_of.p, 0, (size_t)E * sizeof(int), st));
    hipLaunchKernelGGL(k_mark_row_starts, dim3(blocks_for(V)), dim3(kThreads), 0, st, d_ptr, V, o.row_of.p);
    if ((rc = prims::scan<prims::OpMax, false>(o.row_of.p, o.row_of.p, E, tmp.scan, st))) return rc;
    hipLaunchKernelGGL(k_gather_sorted, dim3(blocks_for(E)), dim3(kThreads), 0, st, o.eperm.p, o.row_of.p, d_idx, E, o.row_s.p, o.idx_s.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));   // key / val go out of scope
    return GNNAGG_OK;
}

// gflag / gid (exclusive prefix sum of gflag) of the groups of the sorted edge list; *G_out groups
int cut_groups(const Sorted &s, int E, int ng, hipStream_t st, DevBuf<int> &gflag, DevBuf<int> &gid, Tmp &tmp, int *G_out)
{
    int rc;
    DevBuf<int> ss;
    if ((rc = ss.alloc((size_t)E)) || (rc = gflag.alloc((size_t)E)) || (rc = gid.alloc((size_t)E))) return rc;
    hipLaunchKernelGGL(k_subrow_starts, dim3(blocks_for(E)), dim3(kThreads), 0, st, s.key_s.p, s.row_s.p, E, ss.p);
    if ((rc = prims::scan<prims::OpMax, false>(ss.p, ss.p, E, tmp.scan, st))) return rc;
    hipLaunchKernelGGL(k_group_flags, dim3(blocks_for(E)), dim3(kThreads), 0, st, ss.p, E, ng, gflag.p);
    if ((rc = prims::scan<prims::OpSum, true>(gflag.p, gid.p, E, tmp.scan, st))) return rc;
    int last[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(&last[0], gid.p + (E - 1), sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(&last[1], gflag.p + (E - 1), sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *G_out = last[0] + last[1];
    return GNNAGG_OK;
}

}  // namespace

int gpu_max_col(const int *d_idx, int E, hipStream_t st, int *max_col)
{
    *max_col = 0;
    if (E <= 0) return GNNAGG_OK;
    DevBuf<int> out;
    int rc;
    if ((rc = out.alloc(1)) || (rc = prims::reduce<prims::OpMax>(d_idx, E, out.p, st))) return rc;
    HIP_TRY(hipMemcpyAsync(max_col, out.p, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (*max_col < 0) *max_col = 0;
    return GNNAGG_OK;
}

int gpu_build_blocked_plan(const int *d_ptr, const int *d_idx, int V, int E, int par_num, int total_cols, int ng, int span_edges,
                           hipStream_t st, GpuBlockedPlan &o)
{
    if (E <= 0 || V <= 0 || par_num < 1 || ng < 1 || par_num > 65535) return fail(GNNAGG_ERR_STATE, "internal: blocked plan of an empty graph");
    const int width = total_cols / par_num;
    if (width < 1) return fail(GNNAGG_ERR_STATE, "internal: more ranges than columns");
    int rc;
    Tmp tmp;
    Sorted s;
    if ((rc = sort_edges_by_range(d_ptr, d_idx, V, E, par_num, width, st, s, tmp))) return rc;
    DevBuf<int> gflag, gid;
    int G = 0;
    if ((rc = cut_groups(s, E, ng, st, gflag, gid, tmp, &G))) return rc;
    o.par_num = par_num; o.total_cols = total_cols; o.ng = ng; o.G = G; o.n_edges = E;
    if ((rc = o.ptr_s.alloc((size_t)G + 1)) || (rc = o.target.alloc((size_t)G))) return rc;
    hipLaunchKernelGGL(k_scatter_groups, dim3(blocks_for(E)), dim3(kThreads), 0, st, gflag.p, gid.p, s.row_s.p, s.key_s.p, E, G, o.ptr_s.p, o.target.p,
                       (unsigned *)nullptr);
    // row -> its groups, ascending: counts, prefix sum, and the group ids stably sorted by row
    DevBuf<int> groups_of, iota;
    if ((rc = groups_of.alloc((size_t)V + 1)) || (rc = iota.alloc((size_t)G)) || (rc = o.rg_ptr.alloc((size_t)V + 1)) || (rc = o.rg_idx.alloc((size_t)G))) return rc;
    HIP_TRY(hipMemsetAsync(groups_of.p, 0, ((size_t)V + 1) * sizeof(int), st));
    hipLaunchKernelGGL(k_count_groups, dim3(blocks_for(G)), dim3(kThreads), 0, st, o.target.p, G, groups_of.p);
    hipLaunchKernelGGL(k_iota, dim3(blocks_for(G)), dim3(kThreads), 0, st, iota.p, G);
    {
        DevBuf<unsigned> tin, tkey;   // (the sort clobbers its input: a copy of the targets; row ids are non-negative)
        if ((rc = tin.alloc((size_t)G)) || (rc = tkey.alloc((size_t)G))) return rc;
        HIP_TRY(hipMemcpyAsync(tin.p, o.target.p, (size_t)G * sizeof(int), hipMemcpyDeviceToDevice, st));
        int bits = 1;
        while ((1L << bits) < (long)V) ++bits;
        if ((rc = prims::sort_pairs(tin.p, tkey.p, iota.p, o.rg_idx.p, G, bits, tmp.counts, tmp.scan, st))) return rc;
        if ((rc = prims::scan<prims::OpSum, true>(groups_of.p, o.rg_ptr.p, (long)V + 1, tmp.scan, st))) return rc;
        HIP_TRY(hipStreamSynchronize(st));   // tin / tkey go out of scope
    }
    if ((rc = o.idx_f.alloc((size_t)E))) return rc;
    hipLaunchKernelGGL(k_flag_ids, dim3(blocks_for(E)), dim3(kThreads), 0, st, s.idx_s.p, gflag.p, s.row_s.p, groups_of.p, E, o.idx_f.p);
    HIP_TRY(hipGetLastError());
    o.eperm.swap(s.eperm);
    // per-group arrays to the host: span cut (a greedy walk), rows with several groups / without any, the schedule queries
    o.h_ptr_s.resize((size_t)G + 1);
    o.h_target.resize((size_t)G);
    std::vector<int> h_groups_of((size_t)V);
    HIP_TRY(hipMemcpyAsync(o.h_ptr_s.data(), o.ptr_s.p, ((size_t)G + 1) * sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(o.h_target.data(), o.target.p, (size_t)G * sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(h_groups_of.data(), groups_of.p, (size_t)V * sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    std::vector<int> span_g(1, 0);
    o.span_cost_prefix.assign(1, 0);
    for (int g = 0; g < G;) {
        const int e0 = o.h_ptr_s[g];
        int h = g + 1;
        while (h < G && o.h_ptr_s[h] - e0 < span_edges) ++h;
        span_g.push_back(h);
        o.span_cost_prefix.push_back((long)o.h_ptr_s[h]);
        g = h;
    }
    o.n_spans = (int)span_g.size() - 1;
    std::vector<int> crows;
    o.h_empty.clear();
    for (int r = 0; r < V; ++r) {
        if (h_groups_of[r] > 1) crows.push_back(r);
        if (h_groups_of[r] == 0) o.h_empty.push_back(r);
    }
    std::stable_sort(crows.begin(), crows.end(), [&](int a, int b) { return h_groups_of[a] > h_groups_of[b]; });
    o.n_crows = (int)crows.size();
    o.n_empty = (int)o.h_empty.size();
    if ((rc = o.span_g.upload(span_g)) || (rc = o.crows.upload(crows)) || (rc = o.empty_rows.upload(o.h_empty))) return rc;
    return GNNAGG_OK;
}

int gpu_build_chain_plan(const int *d_ptr, const int *d_idx, const int *h_ptr, int V, int E, int par_num, int total_cols, int hub_edges,
                         int span_edges, hipStream_t st, GpuChainPlan &o)
{
    o.sorted_rows = false;
    if (E <= 0 || V <= 0 || par_num < 2 || par_num > 65535) return GNNAGG_OK;
    const int width = total_cols / par_num;
    if (width < 1) return GNNAGG_OK;
    int rc;
    Tmp tmp;
    Sorted s;
    if ((rc = sort_edges_by_range(d_ptr, d_idx, V, E, par_num, width, st, s, tmp))) return rc;
    {   // neighbors ascending in every row?  (a row's sub-rows per range, range after range, are then the row in CSR order)
        DevBuf<int> flag;
        if ((rc = flag.alloc(1))) return rc;
        HIP_TRY(hipMemsetAsync(flag.p, 0, sizeof(int), st));
        hipLaunchKernelGGL(k_rows_unsorted, dim3(blocks_for(E)), dim3(kThreads), 0, st, d_idx, s.row_of.p, E, flag.p);
        int unsorted = 0;
        HIP_TRY(hipMemcpyAsync(&unsorted, flag.p, sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (unsorted) return GNNAGG_OK;
    }
    o.sorted_rows = true;
    DevBuf<int> gflag, gid;
    int G0 = 0;
    if ((rc = cut_groups(s, E, 0, st, gflag, gid, tmp, &G0))) return rc;
    DevBuf<int> ptr0, target0;
    DevBuf<unsigned> gkey0;
    if ((rc = ptr0.alloc((size_t)G0 + 1)) || (rc = target0.alloc((size_t)G0)) || (rc = gkey0.alloc((size_t)G0))) return rc;
    hipLaunchKernelGGL(k_scatter_groups, dim3(blocks_for(E)), dim3(kThreads), 0, st, gflag.p, gid.p, s.row_s.p, s.key_s.p, E, G0, ptr0.p, target0.p, gkey0.p);
    // rows with a sub-row of more than hub_edges edges leave the chained launches (whole, on the workgroup-per-row kernel)
    if ((rc = o.hub_mask.alloc((size_t)V))) return rc;
    HIP_TRY(hipMemsetAsync(o.hub_mask.p, 0, (size_t)V, st));
    hipLaunchKernelGGL(k_mark_hub_rows, dim3(blocks_for(G0)), dim3(kThreads), 0, st, ptr0.p, target0.p, G0, hub_edges, o.hub_mask.p);
    // the groups that stay: range by range, the longest sub-rows of a range first (stable: ties keep the row order)
    DevBuf<unsigned> key_len, key_range, key_o, key_r2;
    DevBuf<int> keep, iota, ord1, ord, lens, inv, nptr, kept;
    if ((rc = key_len.alloc((size_t)G0)) || (rc = key_range.alloc((size_t)G0)) || (rc = key_o.alloc((size_t)G0)) || (rc = key_r2.alloc((size_t)G0)) ||
        (rc = keep.alloc((size_t)G0)) || (rc = iota.alloc((size_t)G0)) || (rc = ord1.alloc((size_t)G0)) || (rc = ord.alloc((size_t)G0)) ||
        (rc = lens.alloc((size_t)G0 + 1)) || (rc = inv.alloc((size_t)G0)) || (rc = nptr.alloc((size_t)G0 + 1)) || (rc = kept.alloc(1)))
        return rc;
    hipLaunchKernelGGL(k_chain_keys, dim3(blocks_for(G0)), dim3(kThreads), 0, st, ptr0.p, target0.p, gkey0.p, G0, par_num, o.hub_mask.p, key_len.p, key_range.p, keep.p);
    hipLaunchKernelGGL(k_iota, dim3(blocks_for(G0)), dim3(kThreads), 0, st, iota.p, G0);
    int rbits = 1;
    while ((1 << rbits) < par_num + 1) ++rbits;
    {
        int lbits = 1;   // key_len = ~len: its low lbits bits ascend as len descends, for every len < 2^lbits
        while ((1L << lbits) <= (long)E && lbits < 32) ++lbits;
        if ((rc = prims::sort_pairs(key_len.p, key_o.p, iota.p, ord1.p, G0, lbits, tmp.counts, tmp.scan, st))) return rc;   // longest first (stable)
        hipLaunchKernelGGL(k_gather_u32, dim3(blocks_for(G0)), dim3(kThreads), 0, st, key_range.p, ord1.p, G0, key_r2.p);
        if ((rc = prims::sort_pairs(key_r2.p, key_o.p, ord1.p, ord.p, G0, rbits, tmp.counts, tmp.scan, st))) return rc;      // then by range (stable)
        if ((rc = prims::reduce<prims::OpSum>(keep.p, G0, kept.p, st))) return rc;
    }
    int G = 0;
    HIP_TRY(hipMemcpyAsync(&G, kept.p, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    o.par_num = par_num; o.total_cols = total_cols; o.G = G;
    if (G == 0) return GNNAGG_OK;
    hipLaunchKernelGGL(k_chain_lens, dim3(blocks_for(G0)), dim3(kThreads), 0, st, ord.p, ptr0.p, G0, G, lens.p, inv.p);
    HIP_TRY(hipMemsetAsync(lens.p + G0, 0, sizeof(int), st));
    if ((rc = prims::scan<prims::OpSum, true>(lens.p, nptr.p, (long)G0 + 1, tmp.scan, st))) return rc;
    DevBuf<unsigned> grange;
    if ((rc = o.target.alloc((size_t)G)) || (rc = grange.alloc((size_t)G))) return rc;
    hipLaunchKernelGGL(k_chain_groups, dim3(blocks_for(G)), dim3(kThreads), 0, st, ord.p, target0.p, gkey0.p, G, o.target.p, grange.p);
    o.h_ptr_s.resize((size_t)G + 1);
    o.h_target.resize((size_t)G);
    std::vector<unsigned> h_range((size_t)G);
    std::vector<unsigned char> h_hub((size_t)V);
    HIP_TRY(hipMemcpyAsync(o.h_ptr_s.data(), nptr.p, ((size_t)G + 1) * sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(o.h_target.data(), o.target.p, (size_t)G * sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(h_range.data(), grange.p, (size_t)G * sizeof(unsigned), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(h_hub.data(), o.hub_mask.p, (size_t)V, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const int NE = o.h_ptr_s[(size_t)G];
    o.n_edges = NE;
    if ((rc = o.ptr_s.alloc((size_t)G + 1)) || (rc = o.idx_f.alloc((size_t)NE)) || (rc = o.eperm.alloc((size_t)NE))) return rc;
    HIP_TRY(hipMemcpyAsync(o.ptr_s.p, nptr.p, ((size_t)G + 1) * sizeof(int), hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(k_chain_edges, dim3(blocks_for(E)), dim3(kThreads), 0, st, gflag.p, gid.p, inv.p, ptr0.p, nptr.p, s.idx_s.p, s.eperm.p, E, o.idx_f.p, o.eperm.p);
    HIP_TRY(hipGetLastError());
    // spans never straddle two ranges: every range is a launch of its own
    std::vector<int> span_g(1, 0);
    o.span0.assign(1, 0);
    o.cost.assign((size_t)par_num, std::vector<long>());
    int g = 0;
    for (int p = 0; p < par_num; ++p) {
        std::vector<long> &cost = o.cost[(size_t)p];
        cost.assign(1, 0);
        const long base = g < G ? o.h_ptr_s[g] : 0;
        while (g < G && (int)h_range[g] == p) {
            const int e0 = o.h_ptr_s[g];
            int h = g + 1;
            while (h < G && o.h_ptr_s[h] - e0 < span_edges && (int)h_range[h] == p) ++h;
            span_g.push_back(h);
            cost.push_back((long)o.h_ptr_s[h] - base);
            g = h;
        }
        o.span0.push_back((int)span_g.size() - 1);
    }
    if (g != G) return fail(GNNAGG_ERR_STATE, "internal: chain plan groups out of range order");
    if ((rc = o.span_g.upload(span_g))) return rc;
    // the hub rows, heaviest first, as {beg, end, row, 0} descriptors of the workgroup-per-row kernel
    struct Long { int beg, end, row; };
    std::vector<Long> longs;
    for (int r = 0; r < V; ++r)
        if (h_hub[r]) longs.push_back({h_ptr[r], h_ptr[r + 1], r});
    std::stable_sort(longs.begin(), longs.end(), [](const Long &a, const Long &b) { return a.end - a.beg > b.end - b.beg; });
    std::vector<int> r1;
    for (const Long &l : longs) r1.insert(r1.end(), {l.beg, l.end, l.row, 0});
    o.n_hub = (int)longs.size();
    if (o.n_hub > 0 && (rc = o.r1.upload(r1))) return rc;
    HIP_TRY(hipStreamSynchronize(st));
    return GNNAGG_OK;
}

}  // namespace gnnagg
