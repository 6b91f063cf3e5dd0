// agg_gat.hip -- GAT kernels (fused edge softmax + weighted SpMM: items and balanced plan) and their launchers.
#include "combine.cuh"

namespace gnnagg {

// ------------------------------------------------------------------------------- GAT items
struct GatArgs {
    const int *ptr, *target, *slot, *empty_rows;
    const int *idx;
    const float *att;
    const float *x;
    float *y;
    float *partial, *partial_den, *newval;
    int n_items, n_total, feat, ntiles, nblocks, heads, dhead, remap;
    float slope;
};


// LIST = false: reference aggr_gat (aggr_gat.h:116-164); LIST = true: aggr_gat_fine (:167-205).
template <int VEC, int GROUP, bool LIST>
__global__ __launch_bounds__(block_of<GROUP>()) void k_gat_items(const GatArgs a)
{
    constexpr int ITEMS = block_of<GROUP>() / GROUP;
    const int b = a.remap ? xcd_remap(blockIdx.x, a.nblocks) : (int)blockIdx.x;
    const int tile = b % a.ntiles;
    const int item = (b / a.ntiles) * ITEMS + (int)threadIdx.x / GROUP;
    const int lane = threadIdx.x & (GROUP - 1);
    const int col = (tile * GROUP + lane) * VEC;
    if (item >= a.n_total || col >= a.feat) return;
    const int F = a.feat, H = a.heads;

    if (LIST && item >= a.n_items) {
        const float z[VEC] = {};
        store_pack<VEC>(a.y + (size_t)a.empty_rows[item - a.n_items] * F + col, z);
        return;
    }
    const int beg = a.ptr[item], end = a.ptr[item + 1];
    const int row = (LIST && a.target) ? a.target[item] : item;
    const int h = col / a.dhead;
    const bool head_leader = (col % a.dhead) == 0;
    const int *__restrict__ idx = a.idx;
    const float *__restrict__ att_src = a.att + (size_t)h * 2 + 1;
    const float *__restrict__ xcol = a.x + col;
    const float a_dst = a.att[((size_t)row * H + h) * 2];

    float acc[VEC] = {};
    float den = 0.0f;
    for (int e = beg; e < end; e += kUnroll) {
        int s[kUnroll];
        float as[kUnroll];
        Pack<VEC> xv[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u)
            if (e + u < end) s[u] = idx[e + u];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u)
            if (e + u < end) {
                as[u] = att_src[(size_t)s[u] * H * 2];
                xv[u] = load_pack<VEC>(xcol + (size_t)s[u] * F);
            }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u)
            if (e + u < end) {
                const float w = edge_weight(a_dst, as[u], a.slope);
                if (a.newval && head_leader) a.newval[(size_t)(e + u) * H + h] = w;
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[k] = __builtin_fmaf(xv[u].v[k], w, acc[k]);
                den += w;
            }
    }
    const int sl = (LIST && a.slot) ? a.slot[item] : -1;
    if (sl >= 0) {
        store_pack<VEC>(a.partial + (size_t)sl * F + col, acc);
        if (head_leader) a.partial_den[(size_t)sl * H + h] = den;
        return;
    }
    if (beg == end) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = 0.0f;
    } else if (!LIST || den != 0.0f) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = acc[k] / den;
    }
    store_pack<VEC>(a.y + (size_t)row * F + col, acc);
}

// --------------------------------------------------------------------------- GAT, balanced plan
// The fused edge-softmax + weighted SpMM (reference aggr_gat / aggr_gat_fine, aggr_gat.h:116-205) on the same
// plan as k_gcn_plan: short rows one lane group each, long rows one workgroup per <= 16-chunk segment with the
// numerator AND denominator partials folded in ascending chunk order in LDS, hubs through scratch + k_combine.
template <int VEC, int GROUP, int UNROLL = kUnroll>
__device__ __forceinline__ void chain_edges_gat(float (&acc)[VEC], float &den, int beg, int end, int lane, bool col_ok,
                                                const int *__restrict__ idx, const float *__restrict__ att_src, int H,
                                                float a_dst, float slope, const float *__restrict__ xcol, int F,
                                                float *newval, int h, bool head_leader, const int *__restrict__ eperm = nullptr)
{
    int my_s = 0;
    if (beg + lane < end) my_s = idx[beg + lane];
    for (int cb = beg; cb < end; cb += GROUP) {
        int nx_s = 0;
        if (cb + GROUP + lane < end) nx_s = idx[cb + GROUP + lane];
        const int n = end - cb < GROUP ? end - cb : GROUP;
        for (int j = 0; j < n; j += UNROLL) {
            int s[UNROLL];
            float as[UNROLL];
            Pack<VEC> xv[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) s[u] = __shfl(my_s, j + u, GROUP);
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
                if (j + u < n && col_ok) {
                    as[u] = att_src[(size_t)s[u] * H * 2];
                    xv[u] = load_pack<VEC>(xcol + (size_t)s[u] * F);
                }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
                if (j + u < n && col_ok) {
                    const float w = edge_weight(a_dst, as[u], slope);
                    if (newval && head_leader) newval[(size_t)(eperm ? eperm[cb + j + u] : cb + j + u) * H + h] = w;
#pragma unroll
                    for (int k = 0; k < VEC; ++k) acc[k] = __builtin_fmaf(xv[u].v[k], w, acc[k]);
                    den += w;
                }
        }
        my_s = nx_s;
    }
}

// Single head (the reference's only case, aggr_gat.h:116-205): the weight of an edge is the same for every column, so lane j
// of the group computes it ONCE for edge cb + j -- its own coalesced id, one source-term gather, one exp -- and the group
// shares it with ds_bpermute like the edge values of the GCN chain, instead of every lane gathering and exponentiating
// every edge.  Ids are fetched two windows ahead and source terms one window ahead, so nothing dependent sits on the path;
// the first feature gathers of a window are issued before its weights are needed.  Same values, same order as
// chain_edges_gat (bit-identical results).
template <int VEC, int GROUP, int UNROLL = kUnroll>
__device__ __forceinline__ void chain_edges_gat1(float (&acc)[VEC], float &den, int beg, int end, int lane, bool col_ok,
                                                 const int *__restrict__ idx, const float *__restrict__ att_src, float a_dst,
                                                 float slope, const float *__restrict__ xcol, int F, float *newval,
                                                 bool first_tile, const int *__restrict__ eperm = nullptr)
{
    int s0 = 0, s1 = 0;
    float a0 = 0.0f, a1 = 0.0f;
    if (beg + lane < end) s0 = idx[beg + lane];
    if (beg + GROUP + lane < end) s1 = idx[beg + GROUP + lane];
    if (beg + lane < end) a0 = att_src[(size_t)s0 * 2];
    for (int cb = beg; cb < end; cb += GROUP) {
        int s2 = 0;
        if (cb + 2 * GROUP + lane < end) s2 = idx[cb + 2 * GROUP + lane];
        if (cb + GROUP + lane < end) a1 = att_src[(size_t)s1 * 2];
        const int n = end - cb < GROUP ? end - cb : GROUP;
        float my_w = 0.0f;
        for (int j = 0; j < n; j += UNROLL) {
            int s[UNROLL];
            float w[UNROLL];
            Pack<VEC> xv[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) s[u] = __shfl(s0, j + u, GROUP);
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
                if (j + u < n && col_ok) xv[u] = load_pack<VEC>(xcol + (size_t)s[u] * F);
            if (j == 0) {  // this lane's edge of the window
                my_w = lane < n ? edge_weight(a_dst, a0, slope) : 0.0f;
                if (newval && first_tile && lane < n) newval[eperm ? eperm[cb + lane] : cb + lane] = my_w;
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) w[u] = __shfl(my_w, j + u, GROUP);
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
                if (j + u < n && col_ok) {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) acc[k] = __builtin_fmaf(xv[u].v[k], w[u], acc[k]);
                    den += w[u];
                }
        }
        s0 = s1; s1 = s2; a0 = a1;
    }
}

struct GatPlanArgs {
    const int4 *t0, *t1;
    const int *idx;
    const float *att;
    const float *x;
    float *y;
    float *partial, *partial_den, *newval;
    int n0, n1, feat, ntiles, chunk, heads, dhead, remap, nblocks0, rows_semantics;
    float slope;
    // hubs folded by the last segment workgroup to arrive (hub_count == nullptr: k_combine), as in PlanArgs
    const int *slot_hub, *mrow_ptr, *mrow_id;
    int *hub_count;
    int hub_count_stride;
    unsigned partial_bytes, partial_den_bytes;
    // X / partial addressing and block order as in PlanArgs (agg_gcn.hip); eperm: see GatPlanLaunch
    int xpitch, ppitch, yvec, tile_major, item_blocks;
    long x_tile_stride, p_tile_stride;
    unsigned ptile_bytes;
    const int *eperm;
    // two-pass form (gnnagg_gat_run_part; the row-partitioned step): 1 = first pass, y receives the NUMERATOR and den_io[row, h]
    // the denominator, no division; 2 = last pass, both are added to what the first pass left and the row is divided; 3 = a pass
    // in between (staged halo exchange): both are added, nothing is divided
    int part_mode;
    float *den_io;
    XcdRanges xr;
};

// Last step of a GAT row: softmax division (scaleArray, aggr_gat.h:207-213), or its two-pass form.
template <int VEC>
__device__ __forceinline__ void finish_gat_row(const GatPlanArgs &a, float (&acc)[VEC], float den, int row, int h, bool head_leader,
                                               const float *yold, bool always_divide)
{
    if (a.part_mode == 1) {
        if (head_leader) a.den_io[(size_t)row * a.heads + h] = den;
        return;
    }
    if (a.part_mode >= 2) {
        const Pack<VEC> old = load_pack<VEC>(yold);
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = old.v[k] + acc[k];
        den = a.den_io[(size_t)row * a.heads + h] + den;
        if (a.part_mode == 3) {   // (every lane of the head has read the old value: one wavefront, program order)
            if (head_leader) a.den_io[(size_t)row * a.heads + h] = den;
            return;
        }
    }
    if (den != 0.0f || always_divide) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = acc[k] / den;
    }
}

// GAT counterpart of hub_arrive_and_fold: numerator rows and per-head denominators of the hub's segments, ascending
// slot order, one division at the end (scaleArray, aggr_gat.h:207-213) -- the order of k_combine<.., IS_GAT>.
template <int VEC, int GROUP>
__device__ __forceinline__ void hub_arrive_and_fold_gat(const GatPlanArgs &a, int slot, int tile, int col, bool col_ok, int h,
                                                        bool head_leader, int grp, int lane, float (&acc)[VEC], float den,
                                                        float *stage, float *stage_den)
{
    constexpr int GPB = block_of<GROUP>() / GROUP;
    const int F = a.feat, H = a.heads;
    const __amdgpu_buffer_rsrc_t drsrc = __builtin_amdgcn_make_buffer_rsrc(a.partial_den, 0, (int)a.partial_den_bytes, 0x00020000);
    if (grp == 0 && col_ok) {
        store_pack_wt<VEC>(a.partial, a.partial_bytes, (size_t)slot * F + col, acc);
        if (head_leader) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(den), drsrc, (int)(((size_t)slot * H + h) * 4), 0, 16);
    }
    // publication protocol: see hub_arrive_and_fold (agg_gcn.hip) -- sc1 payload, every storing wave drains, barrier, one
    // agent-scope atomic; the last arriver reads with sc1 loads
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __shared__ int s_hub;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int m = a.slot_hub[slot];
        const int nseg = a.mrow_ptr[m + 1] - a.mrow_ptr[m];
        int *cnt = a.hub_count + (size_t)m * a.hub_count_stride + tile;
        const int old = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == nseg - 1) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_hub = old == nseg - 1 ? m : -1;
    }
    __syncthreads();
    const int m = s_hub;
    if (m < 0) return;
    const int s0 = a.mrow_ptr[m], s1 = a.mrow_ptr[m + 1];
    const int row = a.mrow_id[m];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = 0.0f;
    den = 0.0f;
    for (int sb = s0; sb < s1; sb += kSegChunks) {
        const int nst = s1 - sb < kSegChunks ? s1 - sb : kSegChunks;
        for (int p = grp; p < nst; p += GPB)
            if (col_ok) {
                const Pack<VEC> v = load_pack_sc1<VEC>(a.partial, a.partial_bytes, (size_t)(sb + p) * F + col);
                store_pack<VEC>(&stage[(p * GROUP + lane) * VEC], v.v);
                stage_den[p * GROUP + lane] =
                    __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(drsrc, (int)(((size_t)(sb + p) * H + h) * 4), 0, 16));
            }
        __syncthreads();
        if (grp == 0 && col_ok) {
#pragma unroll
            for (int p = 0; p < kSegChunks; ++p)
                if (p < nst) {
                    const Pack<VEC> v = load_pack<VEC>(&stage[(p * GROUP + lane) * VEC]);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) acc[k] += v.v[k];
                    den += stage_den[p * GROUP + lane];
                }
        }
        __syncthreads();
    }
    if (grp == 0 && col_ok) {
        finish_gat_row<VEC>(a, acc, den, row, h, head_leader, a.y + (size_t)row * F + col, false);
        store_pack<VEC>(a.y + (size_t)row * F + col, acc);
    }
}

// (forcing 6 waves/SIMD -- 80 VGPRs, 5-9 spilled -- changes nothing: 100.8 vs 101.5 us on fig10a, 13.8 vs 13.6 ms on config G)
// (UNROLL: see k_gcn_plan -- 4 on the 32-lane float4 geometry: arxiv-shaped 1 head F = 128, fused balanced 99.8 -> 96.1 us)
template <int VEC, int GROUP, bool SINGLE, int UNROLL = kUnroll>
__global__ __launch_bounds__(block_of<GROUP>()) void k_gat_plan(const GatPlanArgs a)
{
    constexpr int GPB = block_of<GROUP>() / GROUP;
    const int F = a.feat, H = a.heads;
    const int lane = threadIdx.x & (GROUP - 1);
    const int grp = (int)threadIdx.x / GROUP;
    const int nb1 = a.n1 * a.ntiles;
    const bool seg_block = (int)blockIdx.x < nb1;
    int tile, row_or_dest;
    int4 d;
    if (seg_block) {
        tile = (int)blockIdx.x % a.ntiles;
        d = a.t1[(int)blockIdx.x / a.ntiles];
    } else {
        const int b = a.tile_major ? logical_block_tile_major((int)blockIdx.x - nb1, a.item_blocks, a.ntiles, a.xr)
                                   : logical_block((int)blockIdx.x - nb1, a.nblocks0, a.ntiles, a.remap, a.xr);
        if (b < 0) return;
        tile = b % a.ntiles;
        const int item = (b / a.ntiles) * GPB + grp;
        if (item >= a.n0) return;
        d = a.t0[item];
    }
    row_or_dest = d.z;
    const int col = (tile * GROUP + lane) * VEC;
    const bool col_ok = col < F;
    const int h = col_ok ? col / a.dhead : 0;
    const bool head_leader = col_ok && (col % a.dhead) == 0;
    const float *__restrict__ att_src = a.att + (size_t)h * 2 + 1;
    const float *__restrict__ xcol = a.x + (size_t)tile * a.x_tile_stride + lane * VEC;
    if (seg_block) {
        __shared__ float stage[kSegChunks * GROUP * VEC];
        __shared__ float stage_den[kSegChunks * GROUP];
        const int row = d.w;  // destination row of this segment (its attention centre term)
        const float a_dst = a.att[((size_t)row * H + h) * 2];
        const int nch = (d.y - d.x + a.chunk - 1) / a.chunk;
        for (int c = grp; c < nch; c += GPB) {
            float acc[VEC] = {};
            float den = 0.0f;
            const int cb = d.x + c * a.chunk;
            const int ce = cb + a.chunk < d.y ? cb + a.chunk : d.y;
            if constexpr (SINGLE)
                chain_edges_gat1<VEC, GROUP, UNROLL>(acc, den, cb, ce, lane, col_ok, a.idx, att_src, a_dst, a.slope, xcol, F, a.newval,
                                             tile == 0);
            else
                chain_edges_gat<VEC, GROUP, UNROLL>(acc, den, cb, ce, lane, col_ok, a.idx, att_src, H, a_dst, a.slope, xcol, F, a.newval,
                                            h, head_leader);
            store_pack<VEC>(&stage[(c * GROUP + lane) * VEC], acc);
            stage_den[c * GROUP + lane] = den;
        }
        __syncthreads();
        const bool hub_here = row_or_dest < 0 && a.hub_count != nullptr;  // workgroup-uniform
        if (!hub_here && (grp != 0 || !col_ok)) return;
        float acc[VEC] = {};
        float den = 0.0f;
        if (grp == 0 && col_ok) {
#pragma unroll
            for (int c = 0; c < kSegChunks; ++c)
                if (c < nch) {
                    const Pack<VEC> p = load_pack<VEC>(&stage[(c * GROUP + lane) * VEC]);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) acc[k] += p.v[k];
                    den += stage_den[c * GROUP + lane];
                }
        }
        if (hub_here) {
            hub_arrive_and_fold_gat<VEC, GROUP>(a, ~row_or_dest, tile, col, col_ok, h, head_leader, grp, lane, acc, den, stage,
                                                stage_den);
            return;
        }
        if (row_or_dest >= 0) {
            finish_gat_row<VEC>(a, acc, den, row_or_dest, h, head_leader, a.y + (size_t)row_or_dest * F + col, false);
            store_pack<VEC>(a.y + (size_t)row_or_dest * F + col, acc);
        } else {
            store_pack<VEC>(a.partial + (size_t)(~row_or_dest) * F + col, acc);
            if (head_leader) a.partial_den[(size_t)(~row_or_dest) * H + h] = den;
        }
        return;
    }
    const int row = d.w;  // destination row (attention centre term); d.z = where the result goes
    float acc[VEC] = {};
    float den = 0.0f;
    if (d.x < d.y) {
        const float a_dst = a.att[((size_t)row * H + h) * 2];
        if constexpr (SINGLE)
            chain_edges_gat1<VEC, GROUP, UNROLL>(acc, den, d.x, d.y, lane, col_ok, a.idx, att_src, a_dst, a.slope, xcol, a.xpitch, a.newval,
                                         tile == 0, a.eperm);
        else
            chain_edges_gat<VEC, GROUP, UNROLL>(acc, den, d.x, d.y, lane, col_ok, a.idx, att_src, H, a_dst, a.slope, xcol, a.xpitch, a.newval,
                                        h, head_leader, a.eperm);
    }
    if (!col_ok) return;
    if (d.z < 0) {  // one of several groups of its row (source-partitioned order): numerator and denominator to scratch
        const size_t poff = (size_t)(~d.z) * a.ppitch + lane * VEC;
        if (a.ptile_bytes) store_pack_wt<VEC>(a.partial + (size_t)tile * a.p_tile_stride, a.ptile_bytes, poff, acc);
        else store_pack<VEC>(a.partial + (size_t)tile * a.p_tile_stride + poff, acc);
        if (head_leader) a.partial_den[(size_t)(~d.z) * H + h] = den;
        return;
    }
    if (a.part_mode != 0) finish_gat_row<VEC>(a, acc, den, row, h, head_leader, a.y + (size_t)row * F + col, false);
    else if (d.x < d.y) finish_gat_row<VEC>(a, acc, den, row, h, head_leader, nullptr, a.rows_semantics != 0);
    if (a.yvec < VEC || F - col < VEC) store_pack_any<VEC>(a.y + (size_t)row * F + col, acc, F - col, a.yvec);
    else store_pack<VEC>(a.y + (size_t)row * F + col, acc);
}

int launch_gat(const GatLaunch &L, void *stream_v)
{
    hipStream_t stream = (hipStream_t)stream_v;
    if (L.feat <= 0 || L.heads <= 0 || L.feat % L.heads != 0)
        return fail(GNNAGG_ERR_ARG, "GAT needs feat >= 1 and feat % heads == 0");
    const bool list = L.wl.target != nullptr || L.wl.slot != nullptr || L.wl.n_empty > 0;
    const int dhead = L.feat / L.heads;
    const Geometry g = pick_geometry(L.feat, L.x, L.y, L.partial, dhead);
    GatArgs a;
    a.ptr = L.wl.ptr; a.target = L.wl.target; a.slot = L.wl.slot; a.empty_rows = L.wl.empty_rows;
    a.idx = L.idx; a.att = L.att; a.x = L.x; a.y = L.y; a.partial = L.partial; a.partial_den = L.partial_den;
    a.newval = L.newval; a.n_items = L.wl.n_items; a.n_total = L.wl.n_items + L.wl.n_empty; a.feat = L.feat;
    a.ntiles = g.ntiles; a.heads = L.heads; a.dhead = dhead; a.remap = L.xcd_remap; a.slope = L.slope;
    if (a.n_total > 0) {
        const int blk = block_for(g.group);
        a.nblocks = ceil_div(a.n_total, blk / g.group) * g.ntiles;
        if (a.remap && a.nblocks < 64) a.remap = 0;
#define CALL_GAT                                                                                             \
        if (list) hipLaunchKernelGGL((k_gat_items<VEC, GROUP, true>), dim3(a.nblocks), dim3(blk), 0, stream, a);    \
        else      hipLaunchKernelGGL((k_gat_items<VEC, GROUP, false>), dim3(a.nblocks), dim3(blk), 0, stream, a);
        DISPATCH_GEOM(g, CALL_GAT)
#undef CALL_GAT
        HIP_TRY(hipGetLastError());
    }
    if (L.wl.n_mrows > 0) {
        CombineArgs c;
        combine_strides(c, L.feat, g, nullptr);
        c.nn_weight = nullptr; c.nn_out = nullptr; c.nn_cols = 0;
        c.mrow_id = L.wl.mrow_id; c.mrow_ptr = L.wl.mrow_ptr; c.row_ptr = nullptr; c.partial = L.partial;
        c.partial_den = L.partial_den; c.y = L.y; c.n_mrows = L.wl.n_mrows; c.feat = L.feat; c.ntiles = g.ntiles;
        c.heads = L.heads; c.dhead = dhead; c.mean = 0; c.accumulate = 0;
        c.big_rows = L.wl.big_rows; c.n_big = L.heads <= 64 ? L.wl.n_big : 0;
        c.nblocks_small = ceil_div(c.n_mrows, kBlock / g.group) * g.ntiles;
        const int nb_big = c.n_big * g.ntiles;
#define CALL_COMB                                                                                                         \
        hipLaunchKernelGGL((k_combine<VEC, GROUP, false, true, false>), dim3(c.nblocks_small), dim3(kBlock), 0, stream, c);     \
        if (nb_big > 0) hipLaunchKernelGGL((k_combine<VEC, GROUP, false, true, true>), dim3(nb_big), dim3(kBlock), 0, stream, c);
        DISPATCH_GEOM(g, CALL_COMB)
#undef CALL_COMB
        HIP_TRY(hipGetLastError());
    }
    return GNNAGG_OK;
}

int launch_gat_plan(const GatPlanLaunch &L, void *stream_v)
{
    hipStream_t stream = (hipStream_t)stream_v;
    if (L.feat <= 0 || L.heads <= 0 || L.feat % L.heads != 0)
        return fail(GNNAGG_ERR_ARG, "GAT needs feat >= 1 and feat % heads == 0");
    const int dhead = L.feat / L.heads;
    if (L.tile.on && (L.n1 > 0 || dhead % 4 != 0)) return fail(GNNAGG_ERR_STATE, "internal: tiled GAT launch with segments / odd heads");
    const Geometry g = L.tile.on ? Geometry{4, L.tile.tile_w / 4, (L.feat + L.tile.tile_w - 1) / L.tile.tile_w}
                                 : pick_geometry(L.feat, L.x, L.y, L.partial, dhead);
    GatPlanArgs a;
    a.t0 = reinterpret_cast<const int4 *>(L.t0); a.t1 = reinterpret_cast<const int4 *>(L.t1);
    a.idx = L.idx; a.att = L.att; a.x = L.x; a.y = L.y; a.partial = L.partial; a.partial_den = L.partial_den;
    a.newval = L.newval; a.n0 = L.n0; a.n1 = L.n1; a.feat = L.feat; a.ntiles = g.ntiles; a.chunk = L.chunk;
    a.heads = L.heads; a.dhead = dhead; a.remap = L.xcd_remap; a.slope = L.slope; a.rows_semantics = L.rows_semantics;
    a.slot_hub = L.slot_hub; a.mrow_ptr = L.hubs.mrow_ptr; a.mrow_id = L.hubs.mrow_id;
    a.hub_count = L.hub_count; a.hub_count_stride = L.hub_count_stride; a.partial_bytes = a.partial_den_bytes = 0;
    a.xpitch = L.feat; a.ppitch = L.feat; a.x_tile_stride = a.p_tile_stride = g.group * g.vec; a.yvec = g.vec;
    a.tile_major = 0; a.item_blocks = 0; a.ptile_bytes = 0; a.eperm = L.eperm;
    a.part_mode = L.part_mode; a.den_io = L.den_io;
    if (L.part_mode != 0 && (L.tile.on || !L.den_io || L.newval || g.vec != 4 || g.ntiles != 1))
        return fail(GNNAGG_ERR_ARG, "two-pass GAT: 16-byte aligned rows of at most 256 columns on the chunked plan, no newval");
    if (L.tile.on) {
        a.xpitch = L.tile.xpitch; a.x_tile_stride = L.tile.x_tile_stride; a.ppitch = L.tile.ppitch;
        a.p_tile_stride = L.tile.p_tile_stride; a.yvec = L.tile.yvec; a.tile_major = 1;
        const size_t tb = (size_t)L.hubs.n_slots * L.tile.ppitch * sizeof(float);
        a.ptile_bytes = tb < 0x7fffffffULL ? (unsigned)tb : 0u;
    }
    {
        // one column tile only: with several, a head's denominator is written by the tile that holds its first column
        // and the other tiles' last arrivers could not know that store is done
        const size_t pbytes = (size_t)L.hubs.n_slots * L.feat * sizeof(float);
        if (L.hubs.n_mrows == 0 || pbytes >= 0x7fffffffULL || g.ntiles != 1 || L.hub_count_stride < 1) a.hub_count = nullptr;
        else { a.partial_bytes = (unsigned)pbytes; a.partial_den_bytes = (unsigned)((size_t)L.hubs.n_slots * L.heads * sizeof(float)); }
    }
    const bool hubs_in_kernel = a.hub_count != nullptr;
    const int blk = block_for(g.group);
    const int gpb = blk / g.group;
    const int item_blocks = ceil_div(a.n0, gpb);
    a.nblocks0 = item_blocks * g.ntiles;
    a.item_blocks = item_blocks;
    if (a.remap && a.nblocks0 < 64 && !a.tile_major) a.remap = 0;
    int grid0 = a.nblocks0;
    if (a.tile_major) {
        if (!L.t0_cost_prefix) return fail(GNNAGG_ERR_STATE, "internal: tiled launch without item costs");
        grid0 = 8 * fill_xcd_ranges_tile_major(L.t0_cost_prefix, a.n0, gpb, item_blocks, g.ntiles, a.xr);
    } else if (a.remap == 2) {
        if (!L.t0_cost_prefix) a.remap = 1;
        else grid0 = 8 * fill_xcd_ranges(L.t0_cost_prefix, a.n0, gpb, item_blocks, a.xr) * g.ntiles;
    }
    const int grid = a.n1 * g.ntiles + grid0;
    if (grid > 0) {
#define CALL_GP                                                                                              \
        if (a.heads == 1) hipLaunchKernelGGL((k_gat_plan<VEC, GROUP, true>), dim3(grid), dim3(blk), 0, stream, a);   \
        else              hipLaunchKernelGGL((k_gat_plan<VEC, GROUP, false>), dim3(grid), dim3(blk), 0, stream, a);
        const bool u4 = L.unroll == 4 && g.vec == 4 && !L.tile.on;
        if (u4 && g.group == 32) {
            if (a.heads == 1) hipLaunchKernelGGL((k_gat_plan<4, 32, true, 4>), dim3(grid), dim3(blk), 0, stream, a);
            else              hipLaunchKernelGGL((k_gat_plan<4, 32, false, 4>), dim3(grid), dim3(blk), 0, stream, a);
        } else if (u4 && g.group == 64) {
            if (a.heads == 1) hipLaunchKernelGGL((k_gat_plan<4, 64, true, 4>), dim3(grid), dim3(blk), 0, stream, a);
            else              hipLaunchKernelGGL((k_gat_plan<4, 64, false, 4>), dim3(grid), dim3(blk), 0, stream, a);
        } else DISPATCH_GEOM(g, CALL_GP)
#undef CALL_GP
        HIP_TRY(hipGetLastError());
    }
    if (L.hubs.n_mrows > 0 && !hubs_in_kernel && L.part_mode != 0)
        return fail(GNNAGG_ERR_STATE, "two-pass GAT: hub rows are folded inside the plan kernel only (GNNAGG_INKERNEL_COMBINE=0 set?)");
    if (L.hubs.n_mrows > 0 && !hubs_in_kernel) {
        CombineArgs c;
        combine_strides(c, L.feat, g, &L.tile);
        c.mrow_id = L.hubs.mrow_id; c.mrow_ptr = L.hubs.mrow_ptr; c.row_ptr = nullptr; c.partial = L.partial;
        c.partial_den = L.partial_den; c.y = L.y; c.n_mrows = L.hubs.n_mrows; c.feat = L.feat; c.ntiles = g.ntiles;
        c.heads = L.heads; c.dhead = dhead; c.mean = 0; c.accumulate = 0;
        c.nn_weight = nullptr; c.nn_out = nullptr; c.nn_cols = 0;
        c.big_rows = L.hubs.big_rows; c.n_big = L.heads <= 64 ? L.hubs.n_big : 0;
        c.nblocks_small = ceil_div(c.n_mrows, kBlock / g.group) * g.ntiles;
        const int nb_big = c.n_big * g.ntiles;
#define CALL_COMB                                                                                                         \
        hipLaunchKernelGGL((k_combine<VEC, GROUP, false, true, false>), dim3(c.nblocks_small), dim3(kBlock), 0, stream, c);     \
        if (nb_big > 0) hipLaunchKernelGGL((k_combine<VEC, GROUP, false, true, true>), dim3(nb_big), dim3(kBlock), 0, stream, c);
        DISPATCH_GEOM(g, CALL_COMB)
#undef CALL_COMB
        HIP_TRY(hipGetLastError());
    }
    return GNNAGG_OK;
}

// edge values follow a permuted edge list (val_t[e'] = val[perm[e']]): the partitioned orders re-gather them before every run
__global__ void k_permute_val(const int *__restrict__ perm, const float *__restrict__ val, float *__restrict__ val_t, int E)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < E) val_t[e] = val ? val[perm[e]] : 1.0f;
}

int launch_permute_val(const int *perm, const float *val, float *val_t, int E, void *stream_v)
{
    if (E <= 0) return GNNAGG_OK;
    hipLaunchKernelGGL(k_permute_val, dim3(ceil_div(E, 256)), dim3(256), 0, (hipStream_t)stream_v, perm, val, val_t, E);
    HIP_TRY(hipGetLastError());
    return GNNAGG_OK;
}

#ifdef GNNAGG_EXTRAS   // the backward kernels ship in libgnnagg_extras.so only (api_extras.hip)
// ------------------------------------------------------------ backward of the single-head fused aggregation
// Reference: aggr_gat_fine_bwd (aggr_gat.h:222-296, marked "Experiment"; run_bwd :426-434).  With w_e = newval[e] and
// D_r = div[r] saved by the forward pass, p_e = w_e / D_r:
//     dz_e      = p_e * (dout_r . x_s  -  dout_r . out_r) * lrelu'(z_e)        (z_e < 0  <=>  w_e < 1)
//     d_a_b[r,0] = sum_{e in row r} dz_e,   d_a_b[s,1] = sum_{e: src(e) = s} dz_e,   d_feat[s,:] = sum_{e: src = s} p_e dout_r
// The reference scatters with fp32 atomics from a warp per neighbor group (and covers 32 columns, no centre term);
// here every sum is a gather in a fixed order: the per-edge stage below runs on the hub-safe chunked work items of the
// edge-softmax kernels, the row sums reuse add_to_center, and the source-side sums and d_feat run on the TRANSPOSED
// CSR (api.hip builds it once per handle) -- d_feat is then simply the balanced GCN aggregation of dout over A^T with
// edge values p.  Deterministic; compared with orc_gat_bwd within the fp32 tolerance.
struct GatBwdArgs {
    const int *ptr_s, *target, *idx;
    const float *out, *dout, *newval, *div, *x, *rowdot;
    float *dz;
    int n_items, V, F;
    float slope;
};

// rowdot[r] = dout[r,:] . out[r,:]
template <int GROUP, int VEC>
__global__ __launch_bounds__(kBlock) void k_rowdot(const float *__restrict__ a, const float *__restrict__ b,
                                                   float *__restrict__ out, int V, int F)
{
    const int row = blockIdx.x * (kBlock / GROUP) + (int)threadIdx.x / GROUP;
    const int lane = threadIdx.x & (GROUP - 1);
    if (row >= V) return;
    float part = 0.0f;
    for (int c = lane * VEC; c < F; c += GROUP * VEC) {
        const Pack<VEC> x = load_pack<VEC>(a + (size_t)row * F + c), y = load_pack<VEC>(b + (size_t)row * F + c);
#pragma unroll
        for (int k = 0; k < VEC; ++k) part = __builtin_fmaf(x.v[k], y.v[k], part);
    }
    part = group_sum<GROUP>(part);
    if (lane == 0) out[row] = part;
}

// One lane group (32 lanes x VEC columns per tile) per work item (<= chunk edges of one row).  As in the forward chains, lane
// j carries (id, weight) of edge cb + j from one coalesced load (next window prefetched) and 8 edges' gathers are in
// flight.  The 8 per-edge dot products are reduced together: a butterfly that halves the number of values a lane keeps
// at each step (4 + 2 + 1 + 1 + 1 = 9 shuffles for 8 edges instead of 40), after which lane 4u holds edge u's dot and
// finishes it -- eight lanes write eight consecutive dz.
template <int VEC>
__global__ __launch_bounds__(kBlock) void k_gat_bwd_edges(const GatBwdArgs a)
{
    constexpr int GROUP = 32, U = 8;
    const int item = blockIdx.x * (kBlock / GROUP) + (int)threadIdx.x / GROUP;
    const int lane = threadIdx.x & (GROUP - 1);
    if (item >= a.n_items) return;
    const int beg = a.ptr_s[item], end = a.ptr_s[item + 1];
    const int row = a.target[item];
    const int F = a.F;
    const float D = a.div[row];
    const float rd = a.rowdot[row];
    const float *__restrict__ drow = a.dout + (size_t)row * F;
    int my_s = 0;
    float my_w = 0.0f;
    if (beg + lane < end) {
        my_s = a.idx[beg + lane];
        my_w = a.newval[beg + lane];
    }
    for (int cb = beg; cb < end; cb += GROUP) {
        int nx_s = 0;
        float nx_w = 0.0f;
        if (cb + GROUP + lane < end) {
            nx_s = a.idx[cb + GROUP + lane];
            nx_w = a.newval[cb + GROUP + lane];
        }
        const int n = end - cb < GROUP ? end - cb : GROUP;
        for (int j = 0; j < n; j += U) {
            float part[U];
            int s[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                part[u] = 0.0f;
                s[u] = __shfl(my_s, j + u, GROUP);  // past the item's end: id 0, a valid row; its result is dropped
            }
            for (int col = lane * VEC; col < F; col += GROUP * VEC) {
                const Pack<VEC> d = load_pack<VEC>(drow + col);
                Pack<VEC> xv[U];
#pragma unroll
                for (int u = 0; u < U; ++u) xv[u] = load_pack<VEC>(a.x + (size_t)s[u] * F + col);
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int k = 0; k < VEC; ++k) part[u] = __builtin_fmaf(d.v[k], xv[u].v[k], part[u]);
            }
            float q[4], r2[2];
            const bool h16 = (lane & 16) != 0, h8 = (lane & 8) != 0, h4 = (lane & 4) != 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {  // lanes 0-15 keep edges 0-3, lanes 16-31 edges 4-7
                const float recv = __shfl_xor(h16 ? part[u] : part[u + 4], 16, GROUP);
                q[u] = (h16 ? part[u + 4] : part[u]) + recv;
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const float recv = __shfl_xor(h8 ? q[u] : q[u + 2], 8, GROUP);
                r2[u] = (h8 ? q[u + 2] : q[u]) + recv;
            }
            float tsum = (h4 ? r2[1] : r2[0]) + __shfl_xor(h4 ? r2[0] : r2[1], 4, GROUP);
            tsum += __shfl_xor(tsum, 2, GROUP);
            tsum += __shfl_xor(tsum, 1, GROUP);
            const int ul = lane >> 2;  // the edge of the batch whose dot this lane now holds
            const float w = __shfl(my_w, j + ul, GROUP);
            if ((lane & 3) == 0 && j + ul < n) {
                float g = D != 0.0f ? (w / D) * (tsum - rd) : 0.0f;
                if (w < 1.0f) g *= a.slope;
                a.dz[cb + j + ul] = g;
            }
        }
        my_s = nx_s;
        my_w = nx_w;
    }
}

// transposed side: dzT[e'] = dz[perm[e']], valT[e'] = newval[perm[e']] / div[idxT[e']]  (idxT[e'] = destination row of the edge)
__global__ void k_gat_bwd_permute(const int *__restrict__ perm, const int *__restrict__ idx_t, const float *__restrict__ dz,
                                  const float *__restrict__ newval, const float *__restrict__ div, float *__restrict__ dz_t,
                                  float *__restrict__ val_t, int E)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int o = perm[e];
    const float D = div[idx_t[e]];
    dz_t[e] = dz[o];
    val_t[e] = D != 0.0f ? newval[o] / D : 0.0f;
}

// val_t[e'] = val[perm[e']] (1 when the aggregator has implicit unit weights): the edge values of the transposed graph
__global__ void k_interleave2(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) reinterpret_cast<float2 *>(out)[i] = make_float2(a[i], b[i]);
}

int launch_gat_bwd_edges(const GatBwdLaunch &L, void *stream_v)
{
    hipStream_t stream = (hipStream_t)stream_v;
    if (L.V > 0) {
        auto al = [](const void *p) { return ((uintptr_t)p & 15) == 0; };
        if (L.feat % 4 == 0 && al(L.dout) && al(L.out))
            hipLaunchKernelGGL((k_rowdot<16, 4>), dim3(ceil_div(L.V, kBlock / 16)), dim3(kBlock), 0, stream, L.dout, L.out, L.rowdot, L.V, L.feat);
        else
            hipLaunchKernelGGL((k_rowdot<16, 1>), dim3(ceil_div(L.V, kBlock / 16)), dim3(kBlock), 0, stream, L.dout, L.out, L.rowdot, L.V, L.feat);
        HIP_TRY(hipGetLastError());
    }
    if (L.wl.n_items > 0) {
        GatBwdArgs a;
        a.ptr_s = L.wl.ptr; a.target = L.wl.target; a.idx = L.idx; a.out = L.out; a.dout = L.dout; a.newval = L.newval;
        a.div = L.div; a.x = L.x; a.rowdot = L.rowdot; a.dz = L.dz; a.n_items = L.wl.n_items; a.V = L.V; a.F = L.feat;
        a.slope = L.slope;
        auto aligned = [](const void *p) { return ((uintptr_t)p & 15) == 0; };
        const int nb = ceil_div(a.n_items, kBlock / 32);
        if (L.feat % 4 == 0 && aligned(L.x) && aligned(L.dout))
            hipLaunchKernelGGL((k_gat_bwd_edges<4>), dim3(nb), dim3(kBlock), 0, stream, a);
        else
            hipLaunchKernelGGL((k_gat_bwd_edges<1>), dim3(nb), dim3(kBlock), 0, stream, a);
        HIP_TRY(hipGetLastError());
    }
    return GNNAGG_OK;
}

int launch_gat_bwd_permute(const int *perm, const int *idx_t, const float *dz, const float *newval, const float *div, float *dz_t,
                           float *val_t, int E, void *stream_v)
{
    if (E <= 0) return GNNAGG_OK;
    hipLaunchKernelGGL(k_gat_bwd_permute, dim3(ceil_div(E, 256)), dim3(256), 0, (hipStream_t)stream_v, perm, idx_t, dz, newval,
                       div, dz_t, val_t, E);
    HIP_TRY(hipGetLastError());
    return GNNAGG_OK;
}

int launch_interleave2(const float *a, const float *b, float *out, int n, void *stream_v)
{
    if (n <= 0) return GNNAGG_OK;
    hipLaunchKernelGGL(k_interleave2, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream_v, a, b, out, n);
    HIP_TRY(hipGetLastError());
    return GNNAGG_OK;
}

#endif  // GNNAGG_EXTRAS

}  // namespace gnnagg
