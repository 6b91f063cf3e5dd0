// agg_span.hip -- GCN / GraphSAGE aggregation in the 2-D blocked order (source range x column tile) as a SEGMENTED STREAM.
//
// In the blocked order the gathered rows are L2-resident (an XCD walks one (source range, column tile) slice of X at a
// time: tile_w * 4 bytes x rows-of-the-range = 2.5 MB), and the L2 serves row gathers 3-4x faster than the fabric behind
// it does (scripts/micro/gather_ceiling.hip: 24 TB/s for 256-byte segments against 6.3-7.9 TB/s from HBM / Infinity
// Cache).  At that rate the one-work-item-per-lane-group kernels (k_gcn_plan) are bound by their fixed costs instead: a
// (row, range) sub-row has ~20 edges, and every item pays a descriptor fetch, a dependent id fetch and a workgroup slot
// for it (23.5 ms on the reddit-shaped F=602 case at an L2 hit rate of 0.87 and only 0.84 TB/s of fabric traffic).
//
// Here a lane group walks a SPAN: ~kSpanEdges consecutive edges of the permuted (range-major, row-minor) edge list,
// whole groups only.  Ids and values arrive in coalesced windows that ignore group boundaries (the next window is always
// in flight), 8 row gathers are issued before the first FMA, and a group's end is a flag carried in the id's top bit:
// at a flagged edge the accumulators are flushed -- 16 bytes per lane to the group's own partial row (slot = group index,
// so the flushes of a span are one sequential stream), or straight to Y when the group is its row's only one -- and the
// chain restarts from 0.  All spans carry about the same number of edges, so the lane groups of a wavefront and the
// workgroups of an XCD stay in step without any descriptor-level balancing.
//
// Summation order: exactly the groups of the reference's localityNeighborGrouping arrays (graph_schedule.h:156-243),
// each an FMA chain from 0 in list order; a row's group sums are added in ascending group order by k_combine_groups.
// Restated by orc_locality_schedule + orc_gcn_grouped_seg(seg = 0); bit-exact.
#include <type_traits>

#include "kernel_util.cuh"

namespace gnnagg {

#ifndef GNNAGG_PARTIAL_AUX
#define GNNAGG_PARTIAL_AUX 2
#endif
static constexpr int kPartialAux = GNNAGG_PARTIAL_AUX;  // (the flushes cost 0.8 ms of R's 15.05: a build without them runs 14.25 ms)  // partial rows: streaming (nt) stores, see store_pack_wt
// The id (and value) stream is read once per column tile and never again by this workgroup: streaming loads (A/B: GNNAGG_ID_NT)
#ifndef GNNAGG_ID_NT
#define GNNAGG_ID_NT 0
#endif
template <class T>
__device__ __forceinline__ T stream_load(const T *p)
{
#if GNNAGG_ID_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
static constexpr unsigned kLastFlag = 0x80000000u;    // id word: last edge of its group
static constexpr unsigned kDirectFlag = 0x40000000u;  // (with kLastFlag) the group is its row's only group: result goes to Y
static constexpr unsigned kIdMask = 0x3fffffffu;

struct SpanArgs {
    const int *span_g;   // [n_spans + 1]
    const int *ptr_s;    // [G + 1]
    const int *idx_f;    // flagged ids, permuted order
    const float *val_s;  // permuted values or nullptr
    const int *target;   // [G]
    const float *x;
    float *y;
    float *partial;
    int n_spans, span_blocks, feat, ntiles, mean, relu, yvec, xpitch, ppitch;
    int tile0;  // first column tile of this launch (the XCD ranges run over the launch's tiles)
    int xshift_bytes;  // log2 of the X row pitch in bytes when it is a power of two and 32-bit offsets cover a tile image, else -1
    long x_tile_stride, p_tile_stride;
    unsigned ptile_bytes;
    unsigned *probe_sink;
    XcdRanges xr;
};

// FAST_ADDR: every byte offset inside one tile image of X fits 32 bits and ids fit 24 (host-checked): the gather address
// is (uniform tile base) + a 32-bit lane offset -- one 24-bit multiply per gather instead of 64-bit address arithmetic.
#ifdef GNNAGG_SPAN_WAVES  // A/B: ask the register allocator for this many waves per SIMD (default: what 64-80 VGPRs give, 6-8)
#define SPAN_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(GNNAGG_SPAN_WAVES, GNNAGG_SPAN_WAVES)))
#else
#define SPAN_WAVES_ATTR
#endif
// CHAIN (canonical rows mode on the blocked order, gnnagg "rows_blocked"): a launch covers ONE source range; a group is a whole
// (row, range) sub-row; its chain starts from what the row's earlier ranges left in the tiled image of Y (`partial` = Yt[tile][row],
// zeroed before the first range) and goes back there -- for rows whose neighbors are sorted, range after range IS the CSR order, so
// every (row, column) is the reference's one sequential chain (aggr_gcn.h:13-35), bit for bit, with the gathers served by the L2.
// The carry of the NEXT group is requested while the current one is walked (its row comes from a 16-group window of targets).
template <int GROUP, bool IS_MAX, bool HAS_VAL, bool PROBE, bool FAST_ADDR, bool CHAIN = false>
__global__ __launch_bounds__(256) SPAN_WAVES_ATTR void k_gcn_span(const SpanArgs a)
{
    constexpr int VEC = 4, GPB = 256 / GROUP, U = kUnroll;
    const int lane = threadIdx.x & (GROUP - 1);
    const int grp = (int)threadIdx.x / GROUP;
    int tile, sb;
    {
        const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
        if (k >= a.xr.count[xcd]) return;
        const int L = a.xr.first[xcd] + k;
        tile = L / a.span_blocks;
        sb = L - tile * a.span_blocks;
        tile = __builtin_amdgcn_readfirstlane(tile + a.tile0);  // workgroup-uniform: tile bases and buffer resources in SGPRs
    }
    const int s = sb * GPB + grp;
    if (s >= a.n_spans) return;
    const int F = a.feat;
    const int col = (tile * GROUP + lane) * VEC;
    // lanes beyond the last feature column (F = 602: 9 of the 16 lanes of tile 9) neither gather nor store: the address
    // path is what bounds this kernel, and it works per active lane / touched line
    const bool col_ok = col < F;
    int g = a.span_g[s];
    const int g1 = a.span_g[s + 1];
    const int e0 = a.ptr_s[g], e_end = a.ptr_s[g1];
    const float *__restrict__ xtile = a.x + (size_t)tile * a.x_tile_stride;  // wave-uniform
    const float *__restrict__ xcol = xtile + lane * VEC;
    const unsigned lane_off = (unsigned)(lane * VEC), xpitch = (unsigned)a.xpitch;
    float *__restrict__ ptile = a.partial + (size_t)tile * a.p_tile_stride;
    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = IS_MAX ? -INFINITY : 0.0f;
    // CHAIN: targets of the groups gw0 .. gw0 + GROUP - 1 (lane j: group gw0 + j), the current group's row, the next one's row and carry
    int gw0 = g, tw = 0, row_cur = 0, row_n = 0;
    Pack<VEC> cn;
#pragma unroll
    for (int k = 0; k < VEC; ++k) cn.v[k] = 0.0f;
    if constexpr (CHAIN) {
        tw = g + lane < g1 ? a.target[g + lane] : 0;
        row_cur = __shfl(tw, 0, GROUP);
        if (col_ok) {
            const Pack<VEC> c0 = load_pack<VEC>(ptile + (size_t)row_cur * a.ppitch + lane * VEC);
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] = c0.v[k];
        }
        if (g + 1 < g1) {
            row_n = __shfl(tw, 1, GROUP);
            if (col_ok) cn = load_pack<VEC>(ptile + (size_t)row_n * a.ppitch + lane * VEC);
        }
    }
    unsigned sig = 0;
    int gstart = e0;  // first edge of the current group (its length = the degree of a single-group row: mean)
    unsigned my_s = 0;
    float my_w = 1.0f;
    if (e0 + lane < e_end) {
        my_s = (unsigned)stream_load(&a.idx_f[e0 + lane]);
        if (HAS_VAL) my_w = stream_load(&a.val_s[e0 + lane]);
    }
    // one batch of U edges of the window at cb: gathers issued together, then the chain in list order.  JC >= 0: a full window,
    // batch offset known at compile time (no bound checks; the id / value of edge JC + u comes by group_bcast: a DPP move for
    // 16-lane groups); JC == -2: a full window, runtime offset; JC == -1: the last, partial window.
    auto batch = [&](auto jc_tag, int cb, int j, int n) {
        constexpr int JC = decltype(jc_tag)::value;
        constexpr bool FULL = JC != -1;  // -2: a full window with a runtime batch offset (wider lane groups: no DPP form)
        unsigned sr[U];
        float w[U];
        Pack<VEC> xv[U];
        if constexpr (JC >= 0) {
            static_for<U>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                sr[u] = (unsigned)group_bcast<GROUP, JC + u>((int)my_s);
                if (HAS_VAL) w[u] = group_bcast<GROUP, JC + u>(my_w);
            });
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                sr[u] = (unsigned)__shfl((int)my_s, j + u, GROUP);
                if (HAS_VAL) w[u] = __shfl(my_w, j + u, GROUP);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if ((FULL || j + u < n) && col_ok) {
                if constexpr (FAST_ADDR) xv[u] = load_pack<VEC>(xtile + (__umul24(sr[u] & kIdMask, xpitch) + lane_off));
                else xv[u] = load_pack<VEC>(xcol + (size_t)(sr[u] & kIdMask) * a.xpitch);
            }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (FULL || j + u < n) {
                if constexpr (PROBE) {
                    if (col_ok) {
#pragma unroll
                        for (int k = 0; k < VEC; ++k) sig ^= __float_as_uint(xv[u].v[k]);
                    }
                    if (HAS_VAL) sig ^= __float_as_uint(w[u]);
                    continue;
                }
                // Every lane accumulates, also the lanes beyond the last feature column: they did not gather, their xv is whatever the
                // register held, and their acc is never stored (every store below is behind col_ok).  INVARIANT this relies on:
                // `acc` and `xv` of a !col_ok lane never reach another lane or memory -- no shuffle, DPP move or reduction reads them (the
                // only cross-lane traffic in this kernel is ids / values / targets: my_s, my_w, tw), and making those lanes gather instead
                // would cost address cycles for 38 of the 64 lanes of F = 602's last tile.  Pinned by tests/test_gpu_blocked.py::
                // test_blocked_gcn_matches_the_restated_order (F = 602, 100, 30, 33: sum, mean AND max, bit-exact).  With an `if (col_ok)` here the compiler
                // adds unconditionally anyway and then SELECTS -- four v_cndmask per gathered value, a third of this kernel's vector
                // instructions (12.9 per gather instruction, profiles/r05/summary_R_sq.txt); without them R runs 15.40 -> 15.27 ms.
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    if (IS_MAX) {
                        const float p = HAS_VAL ? xv[u].v[k] * w[u] : xv[u].v[k];
                        acc[k] = p > acc[k] ? p : acc[k];
                    } else {
                        // implicit unit weights: fma(x, 1, acc) == acc + x exactly
                        acc[k] = HAS_VAL ? __builtin_fmaf(xv[u].v[k], w[u], acc[k]) : acc[k] + xv[u].v[k];
                    }
                }
                if (CHAIN && (sr[u] & kLastFlag)) {  // the sub-row ends: its chain goes back to Yt, the next row's carry takes over
                    if (col_ok) {
                        if (a.ptile_bytes) store_pack_wt<VEC, kPartialAux>(ptile, a.ptile_bytes, (size_t)row_cur * a.ppitch + lane * VEC, acc);
                        else store_pack<VEC>(ptile + (size_t)row_cur * a.ppitch + lane * VEC, acc);
                    }
                    ++g;
#pragma unroll
                    for (int k = 0; k < VEC; ++k) acc[k] = cn.v[k];
                    row_cur = row_n;
                    if (g + 1 < g1) {
                        if (g + 1 - gw0 >= GROUP) {   // (a window of GROUP groups is ~ a span: rarely more than once per span)
                            gw0 += GROUP;
                            tw = gw0 + lane < g1 ? a.target[gw0 + lane] : 0;
                        }
                        row_n = __shfl(tw, g + 1 - gw0, GROUP);
                        if (col_ok) cn = load_pack<VEC>(ptile + (size_t)row_n * a.ppitch + lane * VEC);
                    }
                } else if (sr[u] & kLastFlag) {  // lane-group uniform: the group ends here
                    const int e_next = cb + j + u + 1;
                    if (sr[u] & kDirectFlag) {
                        const int row = a.target[g];
                        if (a.mean) {
                            const float dg = (float)(e_next - gstart);
#pragma unroll
                            for (int k = 0; k < VEC; ++k) acc[k] = acc[k] / dg;
                        }
                        if (a.relu) relu_pack<VEC>(acc);
                        if (col_ok) store_pack_any<VEC>(a.y + (size_t)row * F + col, acc, F - col, a.yvec);
                    } else if (col_ok) {
                        if (a.ptile_bytes) store_pack_wt<VEC, kPartialAux>(ptile, a.ptile_bytes, (size_t)g * a.ppitch + lane * VEC, acc);
                        else store_pack<VEC>(ptile + (size_t)g * a.ppitch + lane * VEC, acc);
                    }
                    ++g;
                    gstart = e_next;
#pragma unroll
                    for (int k = 0; k < VEC; ++k) acc[k] = IS_MAX ? -INFINITY : 0.0f;
                }
            }
    };
    for (int cb = e0; cb < e_end; cb += GROUP) {
        unsigned nx_s = 0;
        float nx_w = 1.0f;
        if (cb + GROUP + lane < e_end) {
            nx_s = (unsigned)stream_load(&a.idx_f[cb + GROUP + lane]);
            if (HAS_VAL) nx_w = stream_load(&a.val_s[cb + GROUP + lane]);
        }
        const int n = e_end - cb < GROUP ? e_end - cb : GROUP;
        if (n == GROUP) {
            if constexpr (GROUP == 16 && GNNAGG_DPP_SPAN_GCN) {
                static_for<GROUP / U>([&](auto bc) {
                    constexpr int J = decltype(bc)::value * U;
                    batch(std::integral_constant<int, J>{}, cb, J, n);
                });
            } else {
#pragma unroll 1
                for (int j = 0; j < GROUP; j += U) batch(std::integral_constant<int, -2>{}, cb, j, n);
            }
        } else {
#pragma unroll 1
            for (int j = 0; j < n; j += U) batch(std::integral_constant<int, -1>{}, cb, j, n);
        }
        my_s = nx_s;
        my_w = nx_w;
    }
    if constexpr (PROBE) {
        if (sig == 0x9e3779b9u) a.probe_sink[0] = sig;  // practically never: keeps the loads alive
    }
}

// ---------------------------------------------------------------------------------- ordered combine over group lists
struct CombineGroupsArgs {
    const int *crows;    // rows with >= 2 groups, most groups first
    const int *rg_ptr;   // [V + 1] row -> its groups
    const int *rg_idx;   // group ids, ascending per row (= ascending in the schedule's list)
    const int *row_ptr;  // CSR ptr (degrees for mean)
    const float *partial;
    float *y;
    int n_crows, feat, ntiles, mean, relu, yvec, ppitch;
    int tile0;  // first column tile of this launch (ntiles = tiles of the launch)
    long p_tile_stride;
};

// One lane group per (row, column tile): the row's group sums are added in ascending group order -- the deterministic
// counterpart of the reference's atomicAdd (aggr_gcn.h:112).  Group ids come in coalesced windows, 16 partial rows are in
// flight per batch.
template <int GROUP, bool IS_MAX>
__global__ __launch_bounds__(256) void k_combine_groups(const CombineGroupsArgs a)
{
    constexpr int VEC = 4, GPB = 256 / GROUP, CU = 16;
    const int lane = threadIdx.x & (GROUP - 1);
    const int tile = a.tile0 + blockIdx.x % a.ntiles;
    const int i = (blockIdx.x / a.ntiles) * GPB + (int)threadIdx.x / GROUP;
    if (i >= a.n_crows) return;
    const int row = a.crows[i];
    const int F = a.feat;
    const int col = (tile * GROUP + lane) * VEC;
    const bool col_ok = col < F;  // lanes beyond the last column read nothing (their partial columns were never written)
    const int q0 = a.rg_ptr[row], q1 = a.rg_ptr[row + 1];
    const float *__restrict__ ptile = a.partial + (size_t)tile * a.p_tile_stride + lane * VEC;
    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = IS_MAX ? -INFINITY : 0.0f;
    int my_g = q0 + lane < q1 ? a.rg_idx[q0 + lane] : 0;
    for (int qb = q0; qb < q1; qb += GROUP) {
        const int nx_g = qb + GROUP + lane < q1 ? a.rg_idx[qb + GROUP + lane] : 0;
        const int n = q1 - qb < GROUP ? q1 - qb : GROUP;
        for (int j = 0; j < n; j += CU) {
            Pack<VEC> p[CU];
#pragma unroll
            for (int u = 0; u < CU; ++u) {
                const int gsel = __shfl(my_g, j + u, GROUP);
                if (j + u < n && col_ok) p[u] = load_pack<VEC>(ptile + (size_t)gsel * a.ppitch);
            }
#pragma unroll
            for (int u = 0; u < CU; ++u)
                if (j + u < n && col_ok) {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) {
                        if (IS_MAX) acc[k] = p[u].v[k] > acc[k] ? p[u].v[k] : acc[k];
                        else acc[k] += p[u].v[k];
                    }
                }
        }
        my_g = nx_g;
    }
    if (!col_ok) return;
    if (a.mean) {
        const float d = (float)(a.row_ptr[row + 1] - a.row_ptr[row]);
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = acc[k] / d;
    }
    if (a.relu) relu_pack<VEC>(acc);
    store_pack_any<VEC>(a.y + (size_t)row * F + col, acc, F - col, a.yvec);
}

// y[rows[i], :] = 0 (rows without edges: the reference memsets vout, aggr_gcn.h:393)
__global__ __launch_bounds__(256) void k_zero_rows(const int *__restrict__ rows, int n, float *__restrict__ y, int F)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)n * F) return;
    const int r = (int)(i / F);
    y[(size_t)rows[r] * F + (i - (long)r * F)] = 0.0f;
}

// ------------------------------------------------------------------------------------------ GAT on the same stream
// Fused edge softmax + weighted SpMM (reference aggr_gat / aggr_gat_fine, aggr_gat.h:116-205) over spans.  A column tile
// covers HT = tile_w / dhead whole heads (or part of one head: HT = 1).  Lane j of a window computes the weights of edge j once
// per head of the tile (one load of its HT source terms from the compact image of k_tile_att, one exp each) and the group
// shares them; the centre term changes with the GROUP, so the centre terms of the next 2 x GROUP groups ride in two register
// windows (lane j: group gw0 + j, one register per head of the tile), shifted between edge windows -- no dependent load at a
// group boundary.  Flush: numerator to the group's partial row (streaming store), denominator to partial_den[g, h] by the
// lane that holds the head's first column.
struct GatSpanArgs {
    SpanArgs s;
    const float *as_t;    // compact source terms [head group][att_rows][HT] (k_tile_att)
    const float *ac_t;    // compact centre terms, same layout
    float *partial_den;   // [G, H]
    float *newval;        // optional [E, H], CSR edge order
    const int *eperm;     // permuted position -> CSR edge
    int heads, dhead, att_rows;
    float slope;
    // CHAIN (canonical rows mode on the blocked order): per-tile denominator image den_t[tile][den_rows][HT] carried from range to
    // range beside the numerator image Yt (s.partial)
    float *den_t;
    int den_rows;
};

// HT consecutive floats with one load
template <int HT>
__device__ __forceinline__ void load_terms(const float *__restrict__ p, float (&o)[HT])
{
    if constexpr (HT == 1) o[0] = p[0];
    else if constexpr (HT == 2) { const float2 v = *reinterpret_cast<const float2 *>(p); o[0] = v.x; o[1] = v.y; }
    else {
#pragma unroll
        for (int q = 0; q < HT / 4; ++q) {
            const float4 v = *reinterpret_cast<const float4 *>(p + 4 * q);
            o[4 * q] = v.x; o[4 * q + 1] = v.y; o[4 * q + 2] = v.z; o[4 * q + 3] = v.w;
        }
    }
}

// Value of lane SRC of every 16-lane DPP row, written only to the lanes of the DPP banks (4 lanes each) in BANKS; the other
// lanes keep `old`.  Head selection without a compare + select: the lanes of head k of a tile are whole banks.
template <int SRC, int BANKS>
__device__ __forceinline__ float row_bcast_banks(float old, float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), 0x150 + SRC, 0xf, BANKS, false));
}

// (4 waves per SIMD asked for at 1-2 heads per tile: 128 registers -- 9.72 -> 9.47 ms on the reddit-shaped 8 x 32 case; wider
// head counts keep their registers.)  The kernel is bound by VALU issue as much as by the gathers (rocprofv3, reddit-shaped
// 8 x 32: 25 VALU instructions per gathered segment before the round-2 rewrite), so the per-edge path is kept to: one DPP
// row broadcast for the id, one shift-or for the gather offset (SHIFT: line-aligned power-of-two row pitch, 32-bit offsets
// from a wave-uniform tile base), one broadcast per head of the tile for the weight (bank-masked: no select), the FMAs, the
// denominator add and the group-end test.  Everything that happens once per window or once per group stays out of it.
// PROBE: the same id / attention-term loads and tile-row gathers, XOR-consumed; no exp, no chain, no store (gnnagg_gat_probe_gather).
#ifndef GAT_SPAN_U
#define GAT_SPAN_U 16
#endif
#ifndef GAT_SPAN_WAVES
#define GAT_SPAN_WAVES 4
#endif
// CHAIN (gnnagg "rows_blocked", GAT): one launch per source range, a group = a whole (row, range) sub-row; its numerator chain starts
// from what the row's earlier ranges left in Yt[tile][row] and its denominator chain from den_t[tile][row][head] -- both go back
// there at the group's end, the next group's carries are requested while the current one is walked.  For rows whose neighbors are
// sorted, range after range IS the CSR order: every (row, column) numerator and every (row, head) denominator is the one sequential
// chain of aggr_gat (aggr_gat.h:125-163); k_untile_y divides.  A head wider than the tile keeps one denominator copy per tile (the
// tiles of a row run in different workgroups of the same launch: a shared copy would be read after another tile had updated it).
template <int GROUP, int HT, bool SHIFT, bool PROBE, bool CHAIN = false>
__global__ __launch_bounds__(256, (HT <= 2 ? GAT_SPAN_WAVES : 2)) void k_gat_span(const GatSpanArgs A)
{
    const SpanArgs &a = A.s;
    // a whole 16-edge window of gathers in flight (8 per batch: 13.0 ms, 16: 9.7 ms on the reddit-shaped 8 x 32 case)
    constexpr int VEC = 4, GPB = 256 / GROUP, U = GROUP < GAT_SPAN_U ? GROUP : GAT_SPAN_U;
    constexpr bool BANKED = GROUP == 16 && (HT == 2 || HT == 4);  // a head's lanes = whole DPP banks
    const int lane = threadIdx.x & (GROUP - 1);
    const int grp = (int)threadIdx.x / GROUP;
    int tile, sb;
    {
        const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
        if (k >= a.xr.count[xcd]) return;
        const int L = a.xr.first[xcd] + k;
        tile = L / a.span_blocks;
        sb = L - tile * a.span_blocks;
        tile = __builtin_amdgcn_readfirstlane(tile + a.tile0);  // workgroup-uniform: tile bases and buffer resources in SGPRs
    }
    const int s = sb * GPB + grp;
    if (s >= a.n_spans) return;
    const int F = a.feat, H = A.heads;
    const int col = (tile * GROUP + lane) * VEC;
    const bool col_ok = col < F;
    const int h0 = (tile * GROUP * VEC) / A.dhead;            // first head of this tile
    const int hl = (HT > 1 && col_ok) ? col / A.dhead - h0 : 0;  // this lane's head inside the tile
    // which heads of the tile have their first column here (they write newval / partial_den): all of them when a tile
    // holds whole heads; with a head wider than the tile, only the tile where that head starts
    const bool tile_starts_head = HT > 1 || (h0 * A.dhead == tile * GROUP * VEC);
    const bool head_leader = col_ok && (col % A.dhead) == 0;
    int g = a.span_g[s];
    const int g1 = a.span_g[s + 1];
    const int e0 = a.ptr_s[g], e_end = a.ptr_s[g1];
    const float *__restrict__ xtile = a.x + (size_t)tile * a.x_tile_stride;  // wave-uniform
    const float *__restrict__ xcol = xtile + lane * VEC;
    const unsigned lane_boff = (unsigned)(lane * VEC * sizeof(float)), xshift = (unsigned)a.xshift_bytes;
    float *__restrict__ ptile = a.partial + (size_t)tile * a.p_tile_stride;
    const int lane_bit_base = ((int)threadIdx.x & 63) & ~(GROUP - 1);  // first lane of this group inside the wavefront
    // compact attention terms of this tile's head group (k_tile_att): [att_rows][HT]
    const float *__restrict__ as_hg = A.as_t + (size_t)(h0 / HT) * A.att_rows * HT;
    const float *__restrict__ ac_hg = A.ac_t + (size_t)(h0 / HT) * A.att_rows * HT;
    // centre-term windows: lane j holds the centre terms of group gw0 + j (cur) / gw0 + GROUP + j (next), one per tile head.
    // A window of GROUP edges ends at most GROUP groups, so shifting the windows between edge windows keeps every group
    // of the current edge window inside the two (g - gw0 < GROUP at its start).
    int gw0 = g;
    int tw_c, tw_n;
    float aw_c[HT], aw_n[HT];
    auto load_win = [&](int gbase, int &tw, float (&aw)[HT]) {
        tw = gbase + lane < g1 ? a.target[gbase + lane] : 0;
        load_terms<HT>(ac_hg + (size_t)tw * HT, aw);
    };
    load_win(gw0, tw_c, aw_c);
    load_win(gw0 + GROUP, tw_n, aw_n);
    // The weight of an edge is the same for every column of a head, so lane j computes the weights of edge cb + j ONCE per
    // window -- one source-term load and one exp per head of the tile -- and the group shares them like the edge values of
    // the GCN chain (every lane gathering and exponentiating every edge made the kernel VALU-bound: reddit-shaped 8 x 32,
    // 9.97 ms against 5 ms of gathers).  Source terms of window W + 1 are requested while window W is processed.
    unsigned my_s = 0;
    int my_e = 0;
    float as_c[HT];
#pragma unroll
    for (int k = 0; k < HT; ++k) as_c[k] = 0.0f;
    auto load_src_terms = [&](unsigned sw, bool valid, float (&as)[HT]) {
        if (valid) load_terms<HT>(as_hg + (size_t)(sw & kIdMask) * HT, as);
    };
    if (e0 + lane < e_end) {
        my_s = (unsigned)stream_load(&a.idx_f[e0 + lane]);
        if (A.newval) my_e = A.eperm[e0 + lane];
    }
    load_src_terms(my_s, e0 + lane < e_end, as_c);
    float acc[VEC] = {0.f, 0.f, 0.f, 0.f};
    float den = 0.0f;
    unsigned sig = 0;
    // CHAIN: the current group's row, the next one's row and carries (numerator pack, this lane's head's denominator)
    int row_cur = 0, row_n = 0;
    Pack<VEC> cn;
    float dn = 0.0f;
#pragma unroll
    for (int k = 0; k < VEC; ++k) cn.v[k] = 0.0f;
    float *__restrict__ dtile = nullptr;
    const bool den_leader = col_ok && (HT > 1 ? (col % A.dhead) == 0 : lane == 0);
    if constexpr (CHAIN) {
        dtile = A.den_t + (size_t)tile * A.den_rows * HT + hl;
        row_cur = __shfl(tw_c, 0, GROUP);
        if (col_ok) {
            const Pack<VEC> c0 = load_pack<VEC>(ptile + (size_t)row_cur * a.ppitch + lane * VEC);
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] = c0.v[k];
            den = dtile[(size_t)row_cur * HT];
        }
        if (g + 1 < g1) {
            row_n = GROUP > 1 ? __shfl(tw_c, 1, GROUP) : a.target[g + 1];
            if (col_ok) {
                cn = load_pack<VEC>(ptile + (size_t)row_n * a.ppitch + lane * VEC);
                dn = dtile[(size_t)row_n * HT];
            }
        }
    }
    for (int cb = e0; cb < e_end; cb += GROUP) {
        unsigned nx_s = 0;
        int nx_e = 0;
        const bool nx_valid = cb + GROUP + lane < e_end;
        if (nx_valid) {
            nx_s = (unsigned)stream_load(&a.idx_f[cb + GROUP + lane]);
            if (A.newval) nx_e = A.eperm[cb + GROUP + lane];
        }
        const int n = e_end - cb < GROUP ? e_end - cb : GROUP;
        if (n < GROUP) {
            // the span's last window: the lanes beyond it repeat its last edge's source with weight 0 and no flags, so the
            // one unrolled path below serves every window
            const unsigned last = (unsigned)__shfl((int)my_s, n - 1, GROUP) & kIdMask;
            if (lane >= n) my_s = last;
        }
        if (g - gw0 >= GROUP) {  // next centre-term window becomes current; the one after it is requested
            gw0 += GROUP;
            tw_c = tw_n;
#pragma unroll
            for (int k = 0; k < HT; ++k) aw_c[k] = aw_n[k];
            load_win(gw0 + GROUP, tw_n, aw_n);
        }
        // ---- this lane's edge (cb + lane): its group = current group + group ends before it in this window
        float wk[HT];
        if constexpr (PROBE) {
#pragma unroll
            for (int k = 0; k < HT; ++k) {
                wk[k] = 0.0f;
                sig ^= __float_as_uint(as_c[k]) ^ __float_as_uint(aw_c[k]) ^ __float_as_uint(aw_n[k]);
            }
        } else {
            const unsigned long long ends = __ballot((my_s & kLastFlag) != 0);
            const unsigned mine = (unsigned)(ends >> lane_bit_base) & (GROUP >= 32 ? ~0u : ((1u << (GROUP & 31)) - 1u));
            int gi;
            if constexpr (GROUP == 64) gi = __popcll(ends & ((1ull << lane) - 1ull));
            else gi = __popc(mine & ((1u << lane) - 1u));
            gi += g - gw0;  // index into the two windows: < 2 * GROUP
            const int gsel = gi < GROUP ? gi : gi - GROUP;
#pragma unroll
            for (int k = 0; k < HT; ++k) {
                const float c0 = __shfl(aw_c[k], gsel, GROUP), c1 = __shfl(aw_n[k], gsel, GROUP);
                wk[k] = lane < n ? edge_weight(gi < GROUP ? c0 : c1, as_c[k], A.slope) : 0.0f;
            }
            if (A.newval && tile_starts_head && lane < n) {
#pragma unroll
                for (int k = 0; k < HT; ++k)
                    if (h0 + k < H) A.newval[(size_t)my_e * H + h0 + k] = wk[k];
            }
        }
        // one batch of U edges at compile-time offsets: gathers issued together, then the weighted chain
        static_for<GROUP / U>([&](auto bc) {
            constexpr int J = decltype(bc)::value * U;
            Pack<VEC> xv[U];
            static_for<U>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                const unsigned sid = (unsigned)group_bcast<GROUP, J + u>((int)my_s);
                if constexpr (SHIFT)  // the shift drops the flag bits (ids < 2^24, host-checked)
                    xv[u] = load_pack<VEC>(reinterpret_cast<const float *>(reinterpret_cast<const char *>(xtile) + ((sid << xshift) | lane_boff)));
                else
                    xv[u] = load_pack<VEC>(xcol + (size_t)(sid & kIdMask) * a.xpitch);
            });
            static_for<U>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                // (the id word is fetched from its lane again instead of being kept: 16 registers fewer per lane, and
                // registers -- 3 vs 4 waves per SIMD -- are what limits the gathers in flight here)
                const unsigned sru = (unsigned)group_bcast<GROUP, J + u>((int)my_s);
                if constexpr (PROBE) {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) sig ^= __float_as_uint(xv[u].v[k]);
                    if (sru & kLastFlag) ++g;  // (the centre-term windows advance as in the real run)
                    return;
                }
                float w = group_bcast<GROUP, J + u>(wk[0]);
                if constexpr (BANKED) {
                    static_for<HT - 1>([&](auto kc) {
                        constexpr int k = decltype(kc)::value + 1;
                        constexpr int banks = HT == 2 ? 0xC : (1 << k);
                        w = row_bcast_banks<J + u, banks>(w, wk[k]);
                    });
                } else {
                    static_for<HT - 1>([&](auto kc) {
                        constexpr int k = decltype(kc)::value + 1;
                        const float v = group_bcast<GROUP, J + u>(wk[k]);
                        w = hl == k ? v : w;
                    });
                }
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[k] = __builtin_fmaf(xv[u].v[k], w, acc[k]);
                den += w;
                if (CHAIN && (sru & kLastFlag)) {  // the sub-row ends: its chains go back to Yt / den_t, the next row's carries take over
                    if (col_ok) {
                        if (a.ptile_bytes) store_pack_wt<VEC, kPartialAux>(ptile, a.ptile_bytes, (size_t)row_cur * a.ppitch + lane * VEC, acc);
                        else store_pack<VEC>(ptile + (size_t)row_cur * a.ppitch + lane * VEC, acc);
                        if (den_leader) dtile[(size_t)row_cur * HT] = den;
                    }
                    ++g;
#pragma unroll
                    for (int k = 0; k < VEC; ++k) acc[k] = cn.v[k];
                    den = dn;
                    row_cur = row_n;
                    if (g + 1 < g1) {
                        const int gi = g + 1 - gw0;  // <= 2 * GROUP (the windows shift between edge windows only)
                        const int r0 = __shfl(tw_c, gi & (GROUP - 1), GROUP), r1 = __shfl(tw_n, gi & (GROUP - 1), GROUP);
                        row_n = gi < GROUP ? r0 : gi < 2 * GROUP ? r1 : a.target[g + 1];
                        if (col_ok) {
                            cn = load_pack<VEC>(ptile + (size_t)row_n * a.ppitch + lane * VEC);
                            dn = dtile[(size_t)row_n * HT];
                        }
                    }
                } else if (sru & kLastFlag) {  // lane-group uniform: the group ends here
                    if (sru & kDirectFlag) {
                        const int gi = g - gw0;  // < 2 * GROUP
                        const int r0 = __shfl(tw_c, gi & (GROUP - 1), GROUP), r1 = __shfl(tw_n, gi & (GROUP - 1), GROUP);
                        const int row = gi < GROUP ? r0 : r1;
                        if (col_ok) {
                            if (den != 0.0f) {  // scaleArray, aggr_gat.h:207-213
#pragma unroll
                                for (int k = 0; k < VEC; ++k) acc[k] = acc[k] / den;
                            }
                            store_pack_any<VEC>(a.y + (size_t)row * F + col, acc, F - col, a.yvec);
                        }
                    } else {
                        if (a.ptile_bytes) store_pack_wt<VEC, kPartialAux>(ptile, a.ptile_bytes, (size_t)g * a.ppitch + lane * VEC, acc);
                        else store_pack<VEC>(ptile + (size_t)g * a.ppitch + lane * VEC, acc);
                        if (head_leader) A.partial_den[(size_t)g * H + col / A.dhead] = den;
                    }
                    ++g;
                    den = 0.0f;
#pragma unroll
                    for (int k = 0; k < VEC; ++k) acc[k] = 0.0f;
                }
            });
        });
        load_src_terms(nx_s, nx_valid, as_c);  // window W + 1's source terms: in flight across the loop back-edge
        my_s = nx_s;
        my_e = nx_e;
    }
    if constexpr (PROBE) {
        if (sig == 0x9e3779b9u) a.probe_sink[0] = sig;  // practically never: keeps the loads alive
    }
}

struct CombineGroupsGatArgs {
    CombineGroupsArgs c;
    const float *partial_den;
    int heads, dhead;
};

template <int GROUP>
__global__ __launch_bounds__(256) void k_combine_groups_gat(const CombineGroupsGatArgs A)
{
    const CombineGroupsArgs &a = A.c;
    constexpr int VEC = 4, GPB = 256 / GROUP, CU = 16;
    const int lane = threadIdx.x & (GROUP - 1);
    const int tile = a.tile0 + blockIdx.x % a.ntiles;
    const int i = (blockIdx.x / a.ntiles) * GPB + (int)threadIdx.x / GROUP;
    if (i >= a.n_crows) return;
    const int row = a.crows[i];
    const int F = a.feat, H = A.heads;
    const int col = (tile * GROUP + lane) * VEC;
    const bool col_ok = col < F;
    const int h = col_ok ? col / A.dhead : 0;
    const int q0 = a.rg_ptr[row], q1 = a.rg_ptr[row + 1];
    const float *__restrict__ ptile = a.partial + (size_t)tile * a.p_tile_stride + lane * VEC;
    float acc[VEC] = {0.f, 0.f, 0.f, 0.f};
    float den = 0.0f;
    int my_g = q0 + lane < q1 ? a.rg_idx[q0 + lane] : 0;
    for (int qb = q0; qb < q1; qb += GROUP) {
        const int nx_g = qb + GROUP + lane < q1 ? a.rg_idx[qb + GROUP + lane] : 0;
        const int n = q1 - qb < GROUP ? q1 - qb : GROUP;
        for (int j = 0; j < n; j += CU) {
            Pack<VEC> p[CU];
            float pd[CU];
#pragma unroll
            for (int u = 0; u < CU; ++u) {
                const int gsel = __shfl(my_g, j + u, GROUP);
                if (j + u < n) {
                    p[u] = load_pack<VEC>(ptile + (size_t)gsel * a.ppitch);
                    pd[u] = A.partial_den[(size_t)gsel * H + h];
                }
            }
#pragma unroll
            for (int u = 0; u < CU; ++u)
                if (j + u < n) {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) acc[k] += p[u].v[k];
                    den += pd[u];
                }
        }
        my_g = nx_g;
    }
    if (!col_ok) return;
    if (den != 0.0f) {  // scaleArray, aggr_gat.h:207-213
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = acc[k] / den;
    }
    store_pack_any<VEC>(a.y + (size_t)row * F + col, acc, F - col, a.yvec);
}

// Launch plan shared by the GCN and GAT launchers: ONE span launch over all tiles, one ordered combine, the rows without groups.
// (Per-tile / per-chunk launches with the combine on a second stream were measured and lost: scripts/attic/, profiles/r03/overlap_chunks.txt.)
template <class SpanFn, class CombineFn>
static int run_span_plan(const SpanLaunch &L, int ntiles, hipStream_t stream, SpanFn span, CombineFn combine)
{
    if (L.n_spans > 0) { const int rc = span(0, ntiles, stream); if (rc) return rc; }
    if (!L.probe && L.n_crows > 0) { const int rc = combine(0, ntiles, stream); if (rc) return rc; }
    if (!L.probe && L.n_empty > 0) {
        const long total = (long)L.n_empty * L.feat;
        hipLaunchKernelGGL(k_zero_rows, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, L.empty_rows, L.n_empty, L.y, L.feat);
        HIP_TRY(hipGetLastError());
    }
    return GNNAGG_OK;
}

static void fill_span_args(SpanArgs &a, const SpanLaunch &L, int ntiles_total, int group)
{
    a.span_g = L.span_g; a.ptr_s = L.ptr_s; a.idx_f = L.idx_f; a.val_s = L.val_s; a.target = L.target;
    a.x = L.x; a.y = L.y; a.partial = L.partial;
    a.n_spans = L.n_spans; a.feat = L.feat; a.ntiles = ntiles_total; a.mean = L.reduce == GNNAGG_REDUCE_MEAN; a.relu = L.relu;
    a.yvec = L.tile.yvec; a.xpitch = L.tile.xpitch; a.ppitch = L.tile.ppitch;
    a.x_tile_stride = L.tile.x_tile_stride; a.p_tile_stride = L.tile.p_tile_stride;
    const size_t tb = (size_t)L.n_groups * L.tile.ppitch * sizeof(float);
    a.ptile_bytes = (tb < 0x7fffffffULL) ? (unsigned)tb : 0u;
    a.probe_sink = nullptr;
    a.tile0 = 0;
    a.span_blocks = ceil_div(L.n_spans, 256 / group);
    a.xshift_bytes = -1;
    const size_t pitch_b = (size_t)L.tile.xpitch * sizeof(float);
    if (L.x_rows > 0 && L.x_rows < (1 << 24) && (pitch_b & (pitch_b - 1)) == 0 && pitch_b >= 16 && (size_t)L.x_rows * pitch_b < 0xffffffffULL) {
        int sh = 0;
        while (((size_t)1 << sh) < pitch_b) ++sh;
        a.xshift_bytes = sh;
    }
}

static void fill_combine_args(CombineGroupsArgs &c, const SpanLaunch &L)
{
    c.crows = L.crows; c.rg_ptr = L.rg_ptr; c.rg_idx = L.rg_idx; c.row_ptr = L.row_ptr; c.partial = L.partial; c.y = L.y;
    c.n_crows = L.n_crows; c.feat = L.feat; c.mean = L.reduce == GNNAGG_REDUCE_MEAN; c.relu = L.relu;
    c.yvec = L.tile.yvec; c.ppitch = L.tile.ppitch; c.p_tile_stride = L.tile.p_tile_stride;
    c.tile0 = 0; c.ntiles = 1;
}

int launch_gcn_span(const SpanLaunch &L, void *stream_v)
{
    hipStream_t stream = (hipStream_t)stream_v;
    if (L.feat <= 0 || !L.tile.on) return fail(GNNAGG_ERR_STATE, "internal: span launch without a tile spec");
    const int tw = L.tile.tile_w, group = tw / 4;
    if (group != 8 && group != 16 && group != 32 && group != 64) return fail(GNNAGG_ERR_ARG, "unsupported tile width");
    const int ntiles = (L.feat + tw - 1) / tw;
    const bool is_max = L.reduce == GNNAGG_REDUCE_MAX;
    const bool has_val = L.val_s != nullptr;
    if (L.probe && is_max) return fail(GNNAGG_ERR_ARG, "probe: sum/mean only");
    // 32-bit lane offsets inside a tile image / 24-bit ids (FAST_ADDR)
    const bool fast = L.x_rows > 0 && L.x_rows < (1 << 24) && L.tile.xpitch < (1 << 24) &&
                      (size_t)L.x_rows * L.tile.xpitch * sizeof(float) < 0xffffffffULL;
    SpanArgs a0;
    fill_span_args(a0, L, ntiles, group);
    if (L.probe) {
        a0.probe_sink = device_probe_sink();
        if (!a0.probe_sink) return fail(GNNAGG_ERR_HIP, "probe: no sink");
    }
    if (L.chain && (group != 16 || !fast || is_max || L.probe))
        return fail(GNNAGG_ERR_STATE, "internal: chained span launch on a geometry without that kernel");
    auto span = [&](int tile0, int nt, hipStream_t st) -> int {
        SpanArgs a = a0;
        a.tile0 = tile0;
        const int gpb = 256 / group;
        const int grid = 8 * fill_xcd_ranges_tile_major(L.span_cost_prefix, L.n_spans, gpb, a.span_blocks, nt, a.xr);
        if (L.chain) {   // canonical rows mode on the blocked order: one source range per launch, carries through Yt
            if (has_val) hipLaunchKernelGGL((k_gcn_span<16, false, true, false, true, true>), dim3(grid), dim3(256), 0, st, a);
            else         hipLaunchKernelGGL((k_gcn_span<16, false, false, false, true, true>), dim3(grid), dim3(256), 0, st, a);
            HIP_TRY(hipGetLastError());
            return GNNAGG_OK;
        }
#define SPAN_K(G, MAXF, VALF, PROBEF)                                                                                 \
        {                                                                                                               \
            if (fast) hipLaunchKernelGGL((k_gcn_span<G, MAXF, VALF, PROBEF, true>), dim3(grid), dim3(256), 0, st, a);    \
            else      hipLaunchKernelGGL((k_gcn_span<G, MAXF, VALF, PROBEF, false>), dim3(grid), dim3(256), 0, st, a);   \
        }
#define SPAN_CALL(G)                                                                                                    \
        {                                                                                                               \
            if (L.probe) {                                                                                              \
                if (has_val) SPAN_K(G, false, true, true) else SPAN_K(G, false, false, true)                            \
            } else if (is_max) {                                                                                        \
                if (has_val) SPAN_K(G, true, true, false) else SPAN_K(G, true, false, false)                            \
            } else {                                                                                                    \
                if (has_val) SPAN_K(G, false, true, false) else SPAN_K(G, false, false, false)                          \
            }                                                                                                           \
        }
        switch (group) {
            case 8: SPAN_CALL(8) break;
            case 16: SPAN_CALL(16) break;
            case 32: SPAN_CALL(32) break;
            default: SPAN_CALL(64) break;
        }
#undef SPAN_K
#undef SPAN_CALL
        HIP_TRY(hipGetLastError());
        return GNNAGG_OK;
    };
    auto combine = [&](int tile0, int nt, hipStream_t st) -> int {
        CombineGroupsArgs c;
        fill_combine_args(c, L);
        c.tile0 = tile0; c.ntiles = nt;
        const int grid = ceil_div(L.n_crows, 256 / group) * nt;
#define COMB_CALL(G)                                                                                          \
        {                                                                                                     \
            if (is_max) hipLaunchKernelGGL((k_combine_groups<G, true>), dim3(grid), dim3(256), 0, st, c);      \
            else        hipLaunchKernelGGL((k_combine_groups<G, false>), dim3(grid), dim3(256), 0, st, c);     \
        }
        switch (group) {
            case 8: COMB_CALL(8) break;
            case 16: COMB_CALL(16) break;
            case 32: COMB_CALL(32) break;
            default: COMB_CALL(64) break;
        }
#undef COMB_CALL
        HIP_TRY(hipGetLastError());
        return GNNAGG_OK;
    };
    return run_span_plan(L, ntiles, stream, span, combine);
}

int launch_gat_span(const GatSpanLaunch &G, void *stream_v)
{
    hipStream_t stream = (hipStream_t)stream_v;
    const SpanLaunch &L = G.s;
    if (L.feat <= 0 || !L.tile.on || G.heads <= 0 || L.feat % G.heads != 0)
        return fail(GNNAGG_ERR_STATE, "internal: GAT span launch without a tile spec");
    const int tw = L.tile.tile_w, group = tw / 4, dhead = L.feat / G.heads;
    if (group != 8 && group != 16 && group != 32 && group != 64) return fail(GNNAGG_ERR_ARG, "unsupported tile width");
    const int ht = tw >= dhead ? tw / dhead : 1;
    if (!gat_span_tiles(L.feat, G.heads, tw)) return fail(GNNAGG_ERR_STATE, "internal: head width does not tile");
    const int ntiles = (L.feat + tw - 1) / tw;
    GatSpanArgs A0;
    fill_span_args(A0.s, L, ntiles, group);
    A0.s.val_s = nullptr; A0.s.mean = 0; A0.s.relu = 0;
    if (!G.as_t || !G.ac_t || G.att_rows <= 0) return fail(GNNAGG_ERR_STATE, "internal: GAT span launch without the compact attention image");
    if (L.probe) {
        A0.s.probe_sink = device_probe_sink();
        if (!A0.s.probe_sink) return fail(GNNAGG_ERR_HIP, "probe: no sink");
    }
    A0.as_t = G.as_t; A0.ac_t = G.ac_t; A0.att_rows = G.att_rows; A0.partial_den = G.partial_den; A0.newval = G.newval; A0.eperm = G.eperm; A0.heads = G.heads; A0.dhead = dhead;
    A0.slope = G.slope; A0.den_t = G.den_t; A0.den_rows = G.den_rows;
    if (L.chain && (group != 16 || L.probe || G.newval || !G.den_t)) return fail(GNNAGG_ERR_STATE, "internal: chained GAT span launch on a geometry without that kernel");
    auto span = [&](int tile0, int nt, hipStream_t st) -> int {
        GatSpanArgs A = A0;
        A.s.tile0 = tile0;
        const int gpb = 256 / group;
        const int grid = 8 * fill_xcd_ranges_tile_major(L.span_cost_prefix, L.n_spans, gpb, A.s.span_blocks, nt, A.s.xr);
#define GAT_CHAIN_HT(HT_)                                                                                                      \
        {                                                                                                                      \
            if (A.s.xshift_bytes >= 0) hipLaunchKernelGGL((k_gat_span<16, HT_, true, false, true>), dim3(grid), dim3(256), 0, st, A);  \
            else                       hipLaunchKernelGGL((k_gat_span<16, HT_, false, false, true>), dim3(grid), dim3(256), 0, st, A); \
        }
        if (L.chain) {   // canonical rows mode on the blocked order: one source range per launch, carries through Yt / den_t
            switch (ht) {
                case 1: GAT_CHAIN_HT(1); break;
                case 2: GAT_CHAIN_HT(2); break;
                case 4: GAT_CHAIN_HT(4); break;
                default: GAT_CHAIN_HT(8); break;
            }
            HIP_TRY(hipGetLastError());
            return GNNAGG_OK;
        }
#undef GAT_CHAIN_HT
#define GAT_SPAN_HT(G_, HT_)                                                                                                   \
        {                                                                                                                      \
            if (L.probe) {                                                                                                     \
                if (A.s.xshift_bytes >= 0) hipLaunchKernelGGL((k_gat_span<G_, HT_, true, true>), dim3(grid), dim3(256), 0, st, A);  \
                else                       hipLaunchKernelGGL((k_gat_span<G_, HT_, false, true>), dim3(grid), dim3(256), 0, st, A); \
            } else if (A.s.xshift_bytes >= 0) hipLaunchKernelGGL((k_gat_span<G_, HT_, true, false>), dim3(grid), dim3(256), 0, st, A); \
            else                       hipLaunchKernelGGL((k_gat_span<G_, HT_, false, false>), dim3(grid), dim3(256), 0, st, A);      \
        }
#define GAT_SPAN_CALL(G_)                                                      \
        switch (ht) {                                                          \
            case 1: GAT_SPAN_HT(G_, 1); break;                                 \
            case 2: GAT_SPAN_HT(G_, 2); break;                                 \
            case 4: GAT_SPAN_HT(G_, 4); break;                                 \
            default: GAT_SPAN_HT(G_, 8); break;                                \
        }
        switch (group) {
            case 8: GAT_SPAN_CALL(8) break;
            case 16: GAT_SPAN_CALL(16) break;
            case 32: GAT_SPAN_CALL(32) break;
            default: GAT_SPAN_CALL(64) break;
        }
#undef GAT_SPAN_CALL
#undef GAT_SPAN_HT
        HIP_TRY(hipGetLastError());
        return GNNAGG_OK;
    };
    auto combine = [&](int tile0, int nt, hipStream_t st) -> int {
        CombineGroupsGatArgs C;
        fill_combine_args(C.c, L);
        C.c.mean = 0; C.c.relu = 0; C.c.tile0 = tile0; C.c.ntiles = nt;
        C.partial_den = G.partial_den; C.heads = G.heads; C.dhead = dhead;
        const int grid = ceil_div(L.n_crows, 256 / group) * nt;
        switch (group) {
            case 8: hipLaunchKernelGGL((k_combine_groups_gat<8>), dim3(grid), dim3(256), 0, st, C); break;
            case 16: hipLaunchKernelGGL((k_combine_groups_gat<16>), dim3(grid), dim3(256), 0, st, C); break;
            case 32: hipLaunchKernelGGL((k_combine_groups_gat<32>), dim3(grid), dim3(256), 0, st, C); break;
            default: hipLaunchKernelGGL((k_combine_groups_gat<64>), dim3(grid), dim3(256), 0, st, C); break;
        }
        HIP_TRY(hipGetLastError());
        return GNNAGG_OK;
    };
    return run_span_plan(L, ntiles, stream, span, combine);
}

}  // namespace gnnagg
