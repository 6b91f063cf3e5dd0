// kernels.hip -- hand-written CDNA4 (gfx950) kernels of the neighbor-aggregation hot path.
//
// Design (wave64, HBM/L2-bound integer+fp32 gather work; no MFMA -- 0.25 flop/B):
//  * A "lane group" of GROUP = 8/16/32/64 lanes owns one work item (a CSR row, or a chunk of a
//    long row) and one column tile of GROUP*VEC floats; each lane keeps VEC accumulators and
//    loads VEC*4 bytes per neighbor, so a group reads a contiguous GROUP*VEC*4-byte segment of
//    the neighbor's feature row (512 B for F=128: four 128-B lines, one dwordx4 per lane).
//    GROUP < 64 packs several short rows into one wavefront (avg degree of arxiv is 6.9), which
//    is what keeps lanes busy where the reference's warp-per-row scheme idles.
//  * The FMA chain of an item runs in CSR order (bit-exact against the oracle); memory-level
//    parallelism comes from issuing the U=8 neighbor gathers of a batch before the first FMA;
//    (idx,val) of GROUP edges arrive with ONE coalesced load and are broadcast with ds_bpermute.
//  * Long rows are split into several items by the schedule; their partial sums go to a scratch
//    slab and a second kernel adds them in ascending order (deterministic; the reference uses
//    fp32 atomics in arbitrary order, aggr_gcn.h:112).
//  * Workgroup -> item-block mapping is XCD-aware: consecutive item blocks (which share
//    neighbors after the locality reorder) are placed on the same XCD / L2 (8 XCDs, block b runs
//    on XCD b % 8), using the bijective remap so any grid size works.
//  * Wide feature rows (F > GROUP*VEC) are covered by several column tiles; tiles of one item
//    block are adjacent in the remapped block order so they hit the same DRAM pages / L2 lines.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "common.h"

namespace gnnagg {

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess)                                                              \
            return fail(GNNAGG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

static constexpr int kBlock = 256;  // 4 wavefronts
// Aggregation kernels: 8 lane groups per workgroup, at most 4 wavefronts -- 64 threads for 8-lane groups (F <= 32),
// 128 for 16-lane groups, 256 above.  A workgroup's slots are released when its slowest row is done, so narrow
// features (8 or 4 rows per wavefront) want small workgroups: on the arxiv-shaped input F=32 37.3 -> 30.8 us and
// F=64 51.0 -> 48.9 us against 256 threads everywhere; F >= 128 is unchanged (and 64 threads there loses the
// segment path's parallelism: F=256 172 -> 234 us).
template <int GROUP>
constexpr int block_of() { return GROUP * 8 < kBlock ? GROUP * 8 : kBlock; }
static inline int block_for(int group) { return group * 8 < kBlock ? group * 8 : kBlock; }
// neighbor gathers in flight per lane group.  Measured on the arxiv-shaped input: 4 and 8 tie (73.7 / 74.5 us in community
// order, 87.9 / 86.8 us un-reordered), 16 loses (96 us, register pressure), and forcing 8 waves/SIMD with
// __launch_bounds__ spills (137 us): the kernel sits at the memory system's ceiling, not at an occupancy cliff.
static constexpr int kUnroll = 8;

// ---------------------------------------------------------------------------------- helpers
template <int VEC>
struct Pack {
    float v[VEC];
};

template <int VEC>
__device__ __forceinline__ Pack<VEC> load_pack(const float *p)
{
    Pack<VEC> r;
    if constexpr (VEC == 4) {
        const float4 t = *reinterpret_cast<const float4 *>(p);
        r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
    } else if constexpr (VEC == 2) {
        const float2 t = *reinterpret_cast<const float2 *>(p);
        r.v[0] = t.x; r.v[1] = t.y;
    } else {
        r.v[0] = *p;
    }
    return r;
}

// GNNAGG_FLAG_RELU: the activation that follows the aggregation in the 3-layer models (Figure7/our.py:176), applied to the
// finished row in the producing kernel instead of a separate elementwise pass over Y.
template <int VEC>
__device__ __forceinline__ void relu_pack(float (&a)[VEC])
{
#pragma unroll
    for (int k = 0; k < VEC; ++k) a[k] = a[k] > 0.0f ? a[k] : 0.0f;
}

template <int VEC>
__device__ __forceinline__ void store_pack(float *p, const float (&a)[VEC])
{
    if constexpr (VEC == 4) {
        *reinterpret_cast<float4 *>(p) = make_float4(a[0], a[1], a[2], a[3]);
    } else if constexpr (VEC == 2) {
        *reinterpret_cast<float2 *>(p) = make_float2(a[0], a[1]);
    } else {
        *p = a[0];
    }
}

// Output rows are written once and never re-read by this kernel: a write-through (sc1) store leaves the XCD's L2
// to the gathered feature rows instead of parking 87 MB of results in it.  Buffer store so the cache bits can be
// given (aux 16 = sc1); `yoff` is the element offset from `ybase` (callers guarantee the byte offset fits 31 bits).
template <int VEC>
__device__ __forceinline__ void store_pack_wt(float *ybase, unsigned nbytes, size_t yoff, const float (&a)[VEC])
{
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(ybase, 0, (int)nbytes, 0x00020000);
    const int voff = (int)(yoff * sizeof(float));
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    if constexpr (VEC == 4) {
        u4 v = {__float_as_uint(a[0]), __float_as_uint(a[1]), __float_as_uint(a[2]), __float_as_uint(a[3])};
        __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, voff, 0, 16);
    } else if constexpr (VEC == 2) {
        u2 v = {__float_as_uint(a[0]), __float_as_uint(a[1])};
        __builtin_amdgcn_raw_buffer_store_b64(v, rsrc, voff, 0, 16);
    } else {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(a[0]), rsrc, voff, 0, 16);
    }
}

// Device-scope (sc1) load: sees what other XCDs' workgroups wrote with store_pack_wt, whatever this XCD's L2 holds.
template <int VEC>
__device__ __forceinline__ Pack<VEC> load_pack_sc1(const float *base, unsigned nbytes, size_t off)
{
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, (int)nbytes, 0x00020000);
    const int voff = (int)(off * sizeof(float));
    Pack<VEC> r;
    if constexpr (VEC == 4) {
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        const u4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 16);
        r.v[0] = __uint_as_float(v[0]); r.v[1] = __uint_as_float(v[1]); r.v[2] = __uint_as_float(v[2]); r.v[3] = __uint_as_float(v[3]);
    } else if constexpr (VEC == 2) {
        typedef unsigned u2 __attribute__((ext_vector_type(2)));
        const u2 v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, 0, 16);
        r.v[0] = __uint_as_float(v[0]); r.v[1] = __uint_as_float(v[1]);
    } else {
        r.v[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, 0, 16));
    }
    return r;
}

// Block b runs on XCD b % 8 (observed dispatch rule).  Give every XCD a contiguous range of
// logical blocks; bijective for any nb (q = nb/8, r = nb%8: the first r XCDs get q+1 blocks).
__device__ __forceinline__ int xcd_remap(int b, int nb)
{
    const int q = nb >> 3, r = nb & 7;
    const int xcd = b & 7, k = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// Work-balanced XCD ranges: XCD x owns item blocks [first[x], first[x] + count[x]); ranges are cut so
// that every XCD gets about the same number of edges (not the same number of blocks): a locality
// reorder clusters the hub rows, and equal-count ranges would pile them onto one XCD.
struct XcdRanges {
    int first[8];
    int count[8];
};

// remap == 0: identity.  remap == 1: equal-count contiguous ranges.  remap == 2: XcdRanges.
// Returns the logical (item block * ntiles + tile) index, or -1 when this workgroup has no work.
__device__ __forceinline__ int logical_block(int b, int nblocks, int ntiles, int remap, const XcdRanges &xr)
{
    if (remap == 0) return b;
    if (remap == 1) return xcd_remap(b, nblocks);
    const int xcd = b & 7, k = b >> 3;
    const int ib = k / ntiles;
    if (ib >= xr.count[xcd]) return -1;
    return (xr.first[xcd] + ib) * ntiles + (k - ib * ntiles);
}

struct GcnArgs {
    const int *ptr, *target, *slot, *empty_rows;
    const int *row_ptr;
    const int *idx;
    const float *val;
    const float *x;
    float *y;
    float *partial;
    int n_items, n_total, feat, ntiles, nblocks, mean, remap, relu;
    unsigned long long *timer;  // run_clock: per workgroup {first wave start, last wave end, CU id}; null otherwise
    XcdRanges xr;
};

// The FMA (or max) chain of one work item over edges [beg,end) in CSR order.  Lane j of the group fetches
// (idx,val) of edge cb+j with ONE coalesced load per GROUP edges (next window prefetched); each edge's
// pair is broadcast inside the group with ds_bpermute (LDS crossbar, no memory traffic; nontemporal loads of
// this once-streamed metadata were measured 1-5 % SLOWER, and nontemporal feature gathers 30 % slower at an
// unchanged L2 hit rate -- `nt` does not bypass L2 allocation here; neither is used), kUnroll feature
// gathers are issued before the first FMA.  Lanes with col_ok == false still carry metadata.
template <int VEC, int GROUP, bool IS_MAX>
__device__ __forceinline__ void chain_edges(float (&acc)[VEC], int beg, int end, int lane, bool col_ok,
                                            const int *__restrict__ idx, const float *__restrict__ val,
                                            const float *__restrict__ xcol, int F)
{
    int my_s = 0;
    float my_w = 1.0f;
    if (beg + lane < end) {
        my_s = idx[beg + lane];
        if (val) my_w = val[beg + lane];
    }
    for (int cb = beg; cb < end; cb += GROUP) {
        int nx_s = 0;
        float nx_w = 1.0f;
        if (cb + GROUP + lane < end) {
            nx_s = idx[cb + GROUP + lane];
            if (val) nx_w = val[cb + GROUP + lane];
        }
        const int n = end - cb < GROUP ? end - cb : GROUP;
        for (int j = 0; j < n; j += kUnroll) {
            int s[kUnroll];
            float w[kUnroll];
            Pack<VEC> xv[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                s[u] = __shfl(my_s, j + u, GROUP);
                w[u] = __shfl(my_w, j + u, GROUP);
            }
#pragma unroll
            for (int u = 0; u < kUnroll; ++u)
                if (j + u < n && col_ok) xv[u] = load_pack<VEC>(xcol + (size_t)s[u] * F);
#pragma unroll
            for (int u = 0; u < kUnroll; ++u)
                if (j + u < n && col_ok) {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) {
                        if (IS_MAX) {
                            const float p = xv[u].v[k] * w[u];
                            acc[k] = p > acc[k] ? p : acc[k];
                        } else {
                            acc[k] = __builtin_fmaf(xv[u].v[k], w[u], acc[k]);
                        }
                    }
                }
        }
        my_s = nx_s;
        my_w = nx_w;
    }
}

// ------------------------------------------------------------------------- GCN / SAGE items
// LIST = false: item g is CSR row g (reference aggr_gcn, aggr_gcn.h:5-36): the `scheduled = 0` path.
// LIST = true : item g is a group of the schedule (reference aggr_gcn_target, aggr_gcn.h:78-114).
template <int VEC, int GROUP, bool IS_MAX, bool LIST>
__device__ __forceinline__ void gcn_items_body(const GcnArgs &a)
{
    constexpr int ITEMS = block_of<GROUP>() / GROUP;
    const int b = logical_block(blockIdx.x, a.nblocks, a.ntiles, a.remap, a.xr);
    if (b < 0) return;
    const int tile = b % a.ntiles;
    const int item = (b / a.ntiles) * ITEMS + (int)threadIdx.x / GROUP;
    const int lane = threadIdx.x & (GROUP - 1);
    const int col = (tile * GROUP + lane) * VEC;
    if (item >= a.n_total) return;
    const bool col_ok = col < a.feat;  // out-of-range column lanes stay alive: they carry (idx,val) for the broadcast
    const int F = a.feat;

    if (LIST && item >= a.n_items) {  // rows without any group: the reference memsets vout (:393)
        const float z[VEC] = {};
        if (col_ok) store_pack<VEC>(a.y + (size_t)a.empty_rows[item - a.n_items] * F + col, z);
        return;
    }
    const int beg = a.ptr[item], end = a.ptr[item + 1];
    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = IS_MAX ? -INFINITY : 0.0f;
    chain_edges<VEC, GROUP, IS_MAX>(acc, beg, end, lane, col_ok, a.idx, a.val, a.x + col, F);
    if (!col_ok) return;
    const int sl = (LIST && a.slot) ? a.slot[item] : -1;
    if (sl >= 0) {
        store_pack<VEC>(a.partial + (size_t)sl * F + col, acc);
        return;
    }
    const int row = (LIST && a.target) ? a.target[item] : item;
    if (beg == end) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = 0.0f;
    } else if (a.mean) {
        const float d = (float)(end - beg);
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = acc[k] / d;
    }
    if (a.relu) relu_pack<VEC>(acc);
    store_pack<VEC>(a.y + (size_t)row * F + col, acc);
}

// Kernel wrapper.  With a.timer set (reference run_clock, aggr_gcn.h:462-489: %globaltimer / %smid per block,
// kernels aggr_gcn_clock :159-201 and aggr_gcn_target_clock :203-248) the first lane of every wavefront stamps the
// constant-rate wall clock (s_memrealtime) before and after the work: timer[3b] = earliest start, timer[3b+1] =
// latest end, timer[3b+2] = hardware CU id (__smid: XCC / SE / CU bits of HW_ID).
template <int VEC, int GROUP, bool IS_MAX, bool LIST>
__global__ __launch_bounds__(block_of<GROUP>()) void k_gcn_items(const GcnArgs a)
{
    if (a.timer && (threadIdx.x & 63) == 0) {
        atomicMin(&a.timer[3 * (size_t)blockIdx.x], (unsigned long long)wall_clock64());
        if (threadIdx.x == 0) a.timer[3 * (size_t)blockIdx.x + 2] = __smid();
    }
    gcn_items_body<VEC, GROUP, IS_MAX, LIST>(a);
    if (a.timer && (threadIdx.x & 63) == 0)
        atomicMax(&a.timer[3 * (size_t)blockIdx.x + 1], (unsigned long long)wall_clock64());
}

__global__ void k_timer_init(unsigned long long *timer, int nblocks)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < nblocks) {
        timer[3 * (size_t)b] = ~0ULL;
        timer[3 * (size_t)b + 1] = 0ULL;
        timer[3 * (size_t)b + 2] = 0ULL;
    }
}

// ------------------------------------------------------------------ GCN / SAGE, balanced plan
// One launch, two kinds of workgroups (heavy ones first in the grid so they never form the tail):
//  * blocks [0, n1*ntiles): one SEGMENT (<= kSegChunks chunks of `chunk` edges) of a long row per
//    workgroup.  Lane group g computes the partial chains of chunks g, g+GPB, ...; the partials meet in
//    LDS and group 0 folds them in ascending chunk order (deterministic; the reference's
//    aggr_gcn_target adds them with fp32 atomics in arbitrary order, aggr_gcn.h:112).  A row that fits one
//    segment is written straight to Y; only rows with several segments (hubs) go through scratch + k_combine.
//  * the remaining blocks: GPB short rows each (deg <= chunk, empty rows included), one lane group per
//    row, descriptor {beg,end,row} fetched with ONE 16-byte load; XCD-aware work-balanced block ranges.
static constexpr int kSegChunks = 16;

struct PlanArgs {
    const int4 *t0;  // {beg, end, row, -}
    const int4 *t1;  // {beg, end, dest (>=0 row, <0 ~scratch slot), -}
    const int *idx;
    const float *val;
    const float *x;
    float *y;
    float *partial;
    int n0, n1, feat, ntiles, chunk, mean, remap, nblocks0;
    int accumulate;  // 1: y += result (rows without edges are left untouched); sum only
    int relu;        // 1: y = max(result, 0)
    int wt;          // 1: write-through (sc1) stores of the short-row results
    unsigned ybytes;
    // hubs (rows with several segments) folded by the last segment workgroup to arrive; hub_count == nullptr: k_combine
    const int *slot_hub, *mrow_ptr, *mrow_id, *row_ptr;
    int *hub_count;
    int hub_count_stride;
    unsigned partial_bytes;
    XcdRanges xr;
};

// Tail of a hub's segment workgroup.  Its segment sum goes to scratch with a write-through (device-scope) store; once
// the store has completed the workgroup bumps the hub's arrival counter, and the workgroup that finds all other
// segments already in folds the scratch rows in ascending slot order (device-scope loads; the order of k_combine --
// so which workgroup arrives last does not matter) and writes the row.  Returns true in that workgroup, with the
// finished row in acc (group 0's lanes).  `stage` = the segment's LDS stage (kSegChunks rows), free by now.
template <int VEC, int GROUP, bool IS_MAX>
__device__ __forceinline__ bool hub_arrive_and_fold(const PlanArgs &a, const int4 d, int tile, int col, bool col_ok, int grp,
                                                    int lane, float (&acc)[VEC], float *stage, int &row_out)
{
    constexpr int GPB = block_of<GROUP>() / GROUP;
    const int F = a.feat;
    const int slot = ~d.z;
    if (grp == 0 && col_ok) store_pack_wt<VEC>(a.partial, a.partial_bytes, (size_t)slot * F + col, acc);
    __builtin_amdgcn_s_waitcnt(0);  // the write-through store has reached the device coherence point
    __shared__ int s_hub;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int m = a.slot_hub[slot];
        const int nseg = a.mrow_ptr[m + 1] - a.mrow_ptr[m];
        int *cnt = a.hub_count + (size_t)m * a.hub_count_stride + tile;
        const int old = atomicAdd(cnt, 1);
        if (old == nseg - 1) atomicExch(cnt, 0);  // everybody is in: ready for the next launch (same path as the adds)
        s_hub = old == nseg - 1 ? m : -1;
    }
    __syncthreads();
    const int m = s_hub;
    if (m < 0) return false;
    const int s0 = a.mrow_ptr[m], s1 = a.mrow_ptr[m + 1];
    const int row = a.mrow_id[m];
    row_out = row;
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = IS_MAX ? -INFINITY : 0.0f;
    for (int sb = s0; sb < s1; sb += kSegChunks) {
        const int nst = s1 - sb < kSegChunks ? s1 - sb : kSegChunks;
        for (int p = grp; p < nst; p += GPB)
            if (col_ok) {
                const Pack<VEC> v = load_pack_sc1<VEC>(a.partial, a.partial_bytes, (size_t)(sb + p) * F + col);
                store_pack<VEC>(&stage[(p * GROUP + lane) * VEC], v.v);
            }
        __syncthreads();
        if (grp == 0 && col_ok) {
#pragma unroll
            for (int p = 0; p < kSegChunks; ++p)
                if (p < nst) {
                    const Pack<VEC> v = load_pack<VEC>(&stage[(p * GROUP + lane) * VEC]);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) {
                        if (IS_MAX) acc[k] = v.v[k] > acc[k] ? v.v[k] : acc[k];
                        else acc[k] += v.v[k];
                    }
                }
        }
        __syncthreads();
    }
    if (grp == 0 && col_ok) {
        if (a.mean) {
            const float dg = (float)(a.row_ptr[row + 1] - a.row_ptr[row]);
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] = acc[k] / dg;
        }
        if (a.accumulate) {
            const Pack<VEC> old = load_pack<VEC>(a.y + (size_t)row * F + col);
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] = old.v[k] + acc[k];
        }
        if (a.relu) relu_pack<VEC>(acc);
        store_pack<VEC>(a.y + (size_t)row * F + col, acc);
    }
    return true;
}

template <int VEC, int GROUP, bool IS_MAX>
__global__ __launch_bounds__(block_of<GROUP>()) void k_gcn_plan(const PlanArgs a)
{
    constexpr int GPB = block_of<GROUP>() / GROUP;
    const int F = a.feat;
    const int lane = threadIdx.x & (GROUP - 1);
    const int grp = (int)threadIdx.x / GROUP;
    const int nb1 = a.n1 * a.ntiles;
    if ((int)blockIdx.x < nb1) {
        __shared__ float stage[kSegChunks * GROUP * VEC];
        const int tile = (int)blockIdx.x % a.ntiles;
        const int4 d = a.t1[(int)blockIdx.x / a.ntiles];
        const int col = (tile * GROUP + lane) * VEC;
        const bool col_ok = col < F;
        const int nch = (d.y - d.x + a.chunk - 1) / a.chunk;
        for (int c = grp; c < nch; c += GPB) {
            float acc[VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] = IS_MAX ? -INFINITY : 0.0f;
            const int cb = d.x + c * a.chunk;
            const int ce = cb + a.chunk < d.y ? cb + a.chunk : d.y;
            chain_edges<VEC, GROUP, IS_MAX>(acc, cb, ce, lane, col_ok, a.idx, a.val, a.x + col, F);
            store_pack<VEC>(&stage[(c * GROUP + lane) * VEC], acc);
        }
        __syncthreads();
        const bool hub_here = d.z < 0 && a.hub_count != nullptr;  // workgroup-uniform
        if (!hub_here && (grp != 0 || !col_ok)) return;
        float acc[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = IS_MAX ? -INFINITY : 0.0f;
        if (grp == 0 && col_ok) {
#pragma unroll
            for (int c = 0; c < kSegChunks; ++c)
                if (c < nch) {
                    const Pack<VEC> p = load_pack<VEC>(&stage[(c * GROUP + lane) * VEC]);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) {
                        if (IS_MAX) acc[k] = p.v[k] > acc[k] ? p.v[k] : acc[k];
                        else acc[k] += p.v[k];
                    }
                }
        }
        if (hub_here) {
            int row;
            hub_arrive_and_fold<VEC, GROUP, IS_MAX>(a, d, tile, col, col_ok, grp, lane, acc, stage, row);
        } else if (d.z >= 0) {
            if (a.mean) {
                const float dg = (float)(d.y - d.x);
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[k] = acc[k] / dg;
            }
            if (a.accumulate) {
                const Pack<VEC> old = load_pack<VEC>(a.y + (size_t)d.z * F + col);
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[k] = old.v[k] + acc[k];
            }
            if (a.relu) relu_pack<VEC>(acc);
            store_pack<VEC>(a.y + (size_t)d.z * F + col, acc);
        } else {
            store_pack<VEC>(a.partial + (size_t)(~d.z) * F + col, acc);
        }
        return;
    }
    const int b = logical_block((int)blockIdx.x - nb1, a.nblocks0, a.ntiles, a.remap, a.xr);
    if (b < 0) return;
    const int tile = b % a.ntiles;
    const int item = (b / a.ntiles) * GPB + grp;
    if (item >= a.n0) return;
    const int col = (tile * GROUP + lane) * VEC;
    const bool col_ok = col < F;
    const int4 d = a.t0[item];
    if (a.accumulate && !a.relu && d.x == d.y) return;  // y += 0
    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = IS_MAX ? -INFINITY : 0.0f;
    chain_edges<VEC, GROUP, IS_MAX>(acc, d.x, d.y, lane, col_ok, a.idx, a.val, a.x + col, F);
    if (!col_ok) return;
    if (a.accumulate) {
        const Pack<VEC> old = load_pack<VEC>(a.y + (size_t)d.z * F + col);
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = old.v[k] + acc[k];
    } else if (d.x == d.y) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = 0.0f;
    } else if (a.mean) {
        const float dg = (float)(d.y - d.x);
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = acc[k] / dg;
    }
    if (a.relu) relu_pack<VEC>(acc);
    if (a.wt) store_pack_wt<VEC>(a.y, a.ybytes, (size_t)d.z * F + col, acc);
    else store_pack<VEC>(a.y + (size_t)d.z * F + col, acc);
}

// ------------------------------------------------- aggregation with the dense combine as its epilogue
// transformed[V,N] = (A . X)[V,K] . W[K,N] in one pass (reference aggr_gcn_nn, aggr_gcn.h:304-359, called by
// run_with_nn :491-499).  The aggregation spreads the K columns of a row over the lanes of a group while the matrix
// cores want rows across lanes, so finished rows meet in LDS: a workgroup aggregates 32 short rows (32/GPB passes
// of the plan kernel's descriptor path), stages them as a [rows][K] tile (pitch K + 4: aligned 16-byte row stores, operand reads two per bank),
// and after ONE barrier its 4 wavefronts each take 16x16 output sub-tiles and run the full-K chain on
// v_mfma_f32_16x16x4_f32 -- f32 in / f32 accumulate, an ascending-k fmaf chain, so the result is bit-for-bit the
// separate GEMM's (and the oracle's).  W (K*N*4 bytes, 16 KB at 128x32) is read through L1/L2, not staged.
// Unlike the reference (partial . W added with atomics per neighbor group) W is applied to the FINAL row: rows that
// are folded from several chunks (segment path, k_combine) get their product from k_dense_rows afterwards.
typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifndef NN_ROWS
#define NN_ROWS 16
#endif
static constexpr int kNnRows = NN_ROWS;  // short rows a workgroup of the fused kernel aggregates and multiplies

// out_row[j] = sum_k yrow[k] * W[k, j] for j = tid, tid + nthreads, ...: one ascending-k fmaf chain per output (the order of
// the MFMA tiles and of the oracle's GEMM); yrow lives in LDS.  For the few rows that are finished one at a time.
__device__ __forceinline__ void row_times_weight(const float *yrow, int K, const float *__restrict__ W, int N, float *out_row,
                                                 int tid, int nthreads)
{
    for (int j = tid; j < N; j += nthreads) {
        float o = 0.0f;
        int k = 0;
        for (; k + 8 <= K; k += 8) {
            float wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) wv[u] = W[(size_t)(k + u) * N + j];
#pragma unroll
            for (int u = 0; u < 8; ++u) o = fmaf(yrow[k + u], wv[u], o);
        }
        for (; k < K; ++k) o = fmaf(yrow[k], W[(size_t)k * N + j], o);
        out_row[j] = o;
    }
}

struct NnArgs {
    const float *weight;  // [K, N] row-major
    float *out;           // [V, N]
    int n_out;
};

// tile: [32][pitch] floats in LDS, columns [K, roundup4(K)) zero; tile_rows[32] = output row or -1.
// Call right after this thread's tile writes: the function holds the barrier that completes the tile, and issues the
// first W operands BEFORE it so their latency overlaps the wait for the slowest wavefront.  KB = MFMAs per operand
// batch; the batch is branch-free (k-quads past K are clamped loads with zeroed operands: 0 * 0 leaves the chain as is).
template <int KB, int ROWS>
__device__ __forceinline__ void tile_times_weight(const float *tile, int pitch, const int *tile_rows, const float *W, int K,
                                                  int Kw, int N, float *out)
{
    const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    const int ncb = (N + 15) >> 4;
    const int Kf = K & ~3;     // k covered by whole quads
    const int kq = lane >> 4;  // k offset inside one MFMA (A: row = lane % 16, k = lane / 16; B: k = lane / 16, col = lane % 16)
    // W operands as buffer loads: the per-lane byte offset in one VGPR (out of range for the padding columns, which
    // then read 0), the k step in an SGPR -- no per-load address registers
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(W), 0, Kw * N * 4, 0x00020000);
    bool synced = false;
    constexpr int RH = ROWS / 16;  // 16-row halves of the tile
#pragma unroll 1
    for (int st = wave; st < RH * ncb; st += (int)blockDim.x >> 6) {
        const int rh = st % RH, cb = st / RH;
        const int col = cb * 16 + (lane & 15);
        const bool cok = col < N;
        const float *arow = tile + (rh * 16 + (lane & 15)) * pitch + kq;
        const int voff = cok ? (kq * N + col) * 4 : 0x7ffffff0;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int k0 = 0; k0 < Kf; k0 += 4 * KB) {
            float av[KB], bv[KB];
#pragma unroll
            for (int t = 0; t < KB; ++t) {
                const int kk = k0 + 4 * t, kc = kk < Kf ? kk : Kf - 4;
                const float b = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wrsrc, voff, kc * N * 4, 0));
                bv[t] = kk < Kf ? b : 0.f;
            }
            if (!synced) {
                __syncthreads();
                synced = true;
            }
#pragma unroll
            for (int t = 0; t < KB; ++t) {
                const int kk = k0 + 4 * t, kc = kk < Kf ? kk : Kf - 4;
                const float v = arow[kc];
                av[t] = kk < Kf ? v : 0.f;
            }
#pragma unroll
            for (int t = 0; t < KB; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], bv[t], acc, 0, 0, 0);
        }
        if (!synced) {
            __syncthreads();
            synced = true;
        }
        if (K != Kf) {  // ragged last quad: k = Kf + kq valid only below K (the tile's padding columns hold 0)
            const float b = (cok && Kf + kq < K) ? W[(size_t)(Kf + kq) * N + col] : 0.f;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[Kf], b, acc, 0, 0, 0);
        }
        // D layout: col = lane % 16, row = 4 * (lane / 16) + reg
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int r = tile_rows[rh * 16 + 4 * kq + v];
            if (r >= 0 && cok) out[(size_t)r * N + col] = acc[v];
        }
    }
    if (!synced) __syncthreads();
}

template <int VEC, int GROUP, bool IS_MAX>
__global__ __launch_bounds__(block_of<GROUP>()) void k_gcn_plan_nn(const PlanArgs a, const NnArgs w)
{
    constexpr int GPB = block_of<GROUP>() / GROUP;
    constexpr int ROWS = GPB > kNnRows ? GPB : kNnRows;  // rows of the tile
    constexpr int PITCH = GROUP * VEC + 4;  // rows stay 16-byte aligned; operand reads (row = lane % 16, k = lane / 16) fall 2 per bank
    constexpr int kTile = ROWS * PITCH, kStage = kSegChunks * GROUP * VEC;
    __shared__ float lds[kTile > kStage ? kTile : kStage];
    __shared__ int tile_rows[ROWS];
    const int F = a.feat;
    const int lane = threadIdx.x & (GROUP - 1);
    const int grp = (int)threadIdx.x / GROUP;
    const int col = lane * VEC;
    const bool col_ok = col < F;
    if ((int)blockIdx.x < a.n1) {  // one segment of a long row: as in k_gcn_plan (ntiles == 1 here)
        const int4 d = a.t1[blockIdx.x];
        const int nch = (d.y - d.x + a.chunk - 1) / a.chunk;
        for (int c = grp; c < nch; c += GPB) {
            float acc[VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] = IS_MAX ? -INFINITY : 0.0f;
            const int cb = d.x + c * a.chunk;
            const int ce = cb + a.chunk < d.y ? cb + a.chunk : d.y;
            chain_edges<VEC, GROUP, IS_MAX>(acc, cb, ce, lane, col_ok, a.idx, a.val, a.x + col, F);
            store_pack<VEC>(&lds[(c * GROUP + lane) * VEC], acc);
        }
        __syncthreads();
        float hub_acc[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) hub_acc[k] = 0.0f;
        if (grp == 0 && col_ok) {
            float acc[VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] = IS_MAX ? -INFINITY : 0.0f;
#pragma unroll
            for (int c = 0; c < kSegChunks; ++c)
                if (c < nch) {
                    const Pack<VEC> p = load_pack<VEC>(&lds[(c * GROUP + lane) * VEC]);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) {
                        if (IS_MAX) acc[k] = p.v[k] > acc[k] ? p.v[k] : acc[k];
                        else acc[k] += p.v[k];
                    }
                }
            if (d.z >= 0) {
                if (a.mean) {
                    const float dg = (float)(d.y - d.x);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) acc[k] = acc[k] / dg;
                }
                store_pack<VEC>(a.y + (size_t)d.z * F + col, acc);
                store_pack<VEC>(&lds[col], acc);  // chunk 0's slot of this lane: read by nobody else
            } else if (a.hub_count == nullptr) {
                store_pack<VEC>(a.partial + (size_t)(~d.z) * F + col, acc);
            } else {
#pragma unroll
                for (int k = 0; k < VEC; ++k) hub_acc[k] = acc[k];
            }
        }
        int row = d.z;
        if (d.z < 0) {  // a hub's segment
            if (a.hub_count == nullptr) return;  // k_combine finishes the row and multiplies it
            if (!hub_arrive_and_fold<VEC, GROUP, IS_MAX>(a, d, 0, col, col_ok, grp, lane, hub_acc, lds, row)) return;
            if (grp == 0 && col_ok) store_pack<VEC>(&lds[col], hub_acc);
        }
        __syncthreads();
        // the row is final: its product, one thread per output column
        row_times_weight(lds, F, w.weight, w.n_out, w.out + (size_t)row * w.n_out, (int)threadIdx.x, block_of<GROUP>());
        return;
    }
    const int b = logical_block((int)blockIdx.x - a.n1, a.nblocks0, 1, a.remap, a.xr);
    if (b < 0) return;
#pragma unroll 1
    for (int pass = 0; pass < ROWS / GPB; ++pass) {
        const int slot = pass * GPB + grp;
        const int item = b * ROWS + slot;
        float acc[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = 0.0f;
        int row = -1;
        if (item < a.n0) {
            const int4 d = a.t0[item];
            row = d.z;
            if (d.x != d.y) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[k] = IS_MAX ? -INFINITY : 0.0f;
                chain_edges<VEC, GROUP, IS_MAX>(acc, d.x, d.y, lane, col_ok, a.idx, a.val, a.x + col, F);
                if (a.mean) {
                    const float dg = (float)(d.y - d.x);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) acc[k] = acc[k] / dg;
                }
            }
            if (col_ok) {
                if (a.wt) store_pack_wt<VEC>(a.y, a.ybytes, (size_t)row * F + col, acc);
                else store_pack<VEC>(a.y + (size_t)row * F + col, acc);
            }
        }
#if !defined(NN_DBG) || NN_DBG != 1
        if (!col_ok) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] = 0.0f;
        }
        store_pack<VEC>(&lds[slot * PITCH + col], acc);
        if (lane == 0) tile_rows[slot] = row;
#endif
    }
#if defined(NN_DBG) && (NN_DBG == 1 || NN_DBG == 2)
    return;
#endif
    tile_times_weight<(GROUP * VEC >= 128 ? 32 : GROUP * VEC / 4), ROWS>(lds, PITCH, tile_rows, w.weight, F, F, w.n_out, w.out);
}

// out[rows[i], :] = Y[rows[i], :] . W for a short list of rows (the rows-mode long rows, finished on the auxiliary
// stream): one workgroup per row, the row staged in LDS, one chain per output column.
__global__ __launch_bounds__(kBlock) void k_dense_rows(const int *__restrict__ rows, const float *__restrict__ Y,
                                                       const float *__restrict__ W, float *__restrict__ out, int K, int N)
{
    extern __shared__ float yrow[];
    const int row = rows[blockIdx.x];
    for (int k = threadIdx.x; k < K; k += kBlock) yrow[k] = Y[(size_t)row * K + k];
    __syncthreads();
    row_times_weight(yrow, K, W, N, out + (size_t)row * N, (int)threadIdx.x, kBlock);
}

// ---------------------------------------------------------- GCN / SAGE, rows mode (canonical order)
// `scheduled = 0` keeps the reference's summation order exactly -- one sequential FMA chain per (row, column)
// in CSR order (aggr_gcn.h:13-35) -- but does not serialise a hub row on one lane group the way a
// warp-per-row kernel does (1.49 ms on the arxiv-shaped input, whose largest row has 15 k edges).
//  * rows of at most `long_deg` edges: one lane group per row -- the short-row path of k_gcn_plan;
//  * longer rows: this kernel, launched on an auxiliary stream so it overlaps the short rows: one 512-thread
//    workgroup per (row, 32-column tile).  The chain itself cannot be split, but the GATHERS can: every lane
//    group fetches the 128-byte tile segments of different neighbors in parallel (8 per group per round,
//    512 edges per round), the segments meet in LDS in edge order, and 32 threads -- one per column -- run the
//    chain from LDS.  The loads of round r+1 are issued before round r is consumed.  Heaviest rows first.
//    (A 15 k-edge row takes ~0.3 ms this way instead of 1.49 ms; the consumer's ~20 cycles per edge bound it.
//    A column-major stage read with b128 was tried and lost to its scattered LDS writes.)
__device__ __forceinline__ float edge_weight(float a_dst, float a_src, float slope);
static constexpr int kLongBlock = 512;

struct RowsLongArgs {
    const int4 *r1;  // {beg, end, row, -}
    const int *idx;
    const float *val;
    const float *x;
    float *y;
    int n1, feat, ntiles32, mean, relu;
    // GAT flavour (reference aggr_gat, aggr_gat.h:116-164): the edge weight is exp(leaky(att[row,h,0] + att[src,h,1]))
    // computed by the gathering lanes; the consumer also runs the denominator chain.  Needs dhead % 32 == 0 so
    // that a 32-column tile lies inside one head.
    const float *att;
    int heads, dhead;
    float slope;
};

static constexpr int kLongGatherThreads = kLongBlock - 64;  // wavefront 0 only consumes
static constexpr int kLongU = 8;                            // neighbors per gather group per round

template <int VEC>
constexpr int long_round_edges() { return (kLongGatherThreads / (32 / VEC)) * kLongU; }

template <int VEC, bool IS_MAX, bool IS_GAT>
__global__ __launch_bounds__(kLongBlock) void k_gcn_rows_long(const RowsLongArgs a)
{
    constexpr int GL = 32 / VEC;                 // lanes of one gather group: GL * VEC = 32 columns = 128 bytes
    constexpr int NG = kLongGatherThreads / GL;  // gather groups per workgroup
    constexpr int U = kLongU;
    constexpr int RE = NG * U;                   // edges per round
    // two stage buffers + two weight buffers [RE]: round r+1 is written while round r is consumed, so one barrier per
    // round orders everything (the buffer written in round r+1 was last read in round r-1).  Stage layout: the values of
    // 4 consecutive edges of one column are contiguous -- element (edge k, column c) at ((k/4) * 32 + (c ^ swz(k/4))) * 4
    // + k % 4 -- so the consumer fetches 4 chain steps with one ds_read_b128, and a gather thread, which holds 8 edges x
    // VEC columns in registers, writes each (column, 4 edges) quad with one ds_write_b128 (a register transpose, no
    // shuffles).  swz(q) = (q >> 1) & 3 XORs the column inside its row of quads: without it the 8 groups of a gather
    // wavefront write 64-byte-strided quads that all fall on the same 8 of the 32 banks (4x slower stores, which also
    // delay the consumer's reads); with it one store instruction covers every bank evenly.
    extern __shared__ float lds[];
    float *stage0 = lds, *stage1 = lds + RE * 32, *wst0 = lds + 2 * RE * 32, *wst1 = wst0 + RE;
    const int F = a.feat;
    const int tile = (int)blockIdx.x % a.ntiles32;
    const int4 d = a.r1[(int)blockIdx.x / a.ntiles32];
    const int nrounds = (d.y - d.x + RE - 1) / RE;
    const int head = IS_GAT ? (tile * 32) / a.dhead : 0;
    if (threadIdx.x < 64) {
        // ---- consumer wavefront: lane c < 32 owns column tile*32 + c and runs its chain from LDS in edge order
        const int c = (int)threadIdx.x;
        const bool consumer = c < 32 && tile * 32 + c < F;
        float acc = IS_MAX ? -INFINITY : 0.0f, den = 0.0f;
        auto step = [&](float xs, float ws) {
            if (IS_MAX) {
                const float p = xs * ws;
                acc = p > acc ? p : acc;
            } else {
                acc = __builtin_fmaf(xs, ws, acc);
                if (IS_GAT) den += ws;
            }
        };
        for (int r = 0; r < nrounds; ++r) {
            __syncthreads();  // round r is staged
            if (!consumer) continue;
            const float *stage = (r & 1) ? stage1 : stage0, *wst = (r & 1) ? wst1 : wst0;
            const int base = d.x + r * RE;
            const int n = d.y - base < RE ? d.y - base : RE;
            // 32 chain steps per batch = 8 + 8 ds_read_b128; the reads of batch b+1 are issued before the steps of batch b
            // (two register sets), so the chain never waits for LDS latency
            auto load32 = [&](float4 (&xs)[8], float4 (&ws)[8], int k) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {  // k % 32 == 0: swz(k/4 + q) == (q >> 1) & 3
                    xs[q] = *reinterpret_cast<const float4 *>(&stage[(((k >> 2) + q) * 32 + (c ^ ((q >> 1) & 3))) * 4]);
                    ws[q] = *reinterpret_cast<const float4 *>(&wst[k + 4 * q]);
                }
            };
            auto steps32 = [&](const float4 (&xs)[8], const float4 (&ws)[8]) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    step(xs[q].x, ws[q].x);
                    step(xs[q].y, ws[q].y);
                    step(xs[q].z, ws[q].z);
                    step(xs[q].w, ws[q].w);
                }
            };
            const int nfull = n & ~31;
            int k0 = 0;
            if (nfull > 0) {
                float4 xa[8], wa[8], xb[8], wb[8];
                load32(xa, wa, 0);
                while (true) {
                    if (k0 + 32 < nfull) load32(xb, wb, k0 + 32);
                    steps32(xa, wa);
                    k0 += 32;
                    if (k0 >= nfull) break;
                    if (k0 + 32 < nfull) load32(xa, wa, k0 + 32);
                    steps32(xb, wb);
                    k0 += 32;
                    if (k0 >= nfull) break;
                }
            }
            for (; k0 < n; ++k0) step(stage[((k0 >> 2) * 32 + (c ^ ((k0 >> 3) & 3))) * 4 + (k0 & 3)], wst[k0]);
        }
        if (consumer) {
            if (IS_GAT) acc = acc / den;  // aggr_gat.h:163 (rows here are never empty)
            else if (a.mean) acc = acc / (float)(d.y - d.x);
            if (!IS_GAT && a.relu) acc = acc > 0.0f ? acc : 0.0f;
            a.y[(size_t)d.z * F + tile * 32 + c] = acc;
        }
        return;
    }
    // ---- gather wavefronts: group g fetches the 128-byte tile segments of edges base + g*U .. +U of every round
    const int t = (int)threadIdx.x - 64;
    const int g = t / GL, lane = t & (GL - 1);
    const int col = tile * 32 + lane * VEC;
    const bool col_ok = col < F;
    const float *__restrict__ xcol = a.x + col;
    const float a_dst = IS_GAT ? a.att[((size_t)d.z * a.heads + head) * 2] : 0.0f;
    // Two rounds of gathers are in flight (register sets A: even rounds, B: odd rounds) on top of the round in LDS, and
    // the neighbor ids / weights are fetched two rounds before their gathers: under the load of the short-row kernel
    // running beside this one every dependent load costs microseconds.  Every lane of a group loads the group's U ids
    // (same addresses: one request each, no LDS shuffles -- the LDS pipe belongs to the stage); lane u < U also carries
    // the u-th edge's weight and writes it to LDS.  Everything is branch-free: edges past the row's end are clamped to
    // the last edge (their stage slots are never read).
    const int mlane = lane < U ? lane : U - 1;
    struct Meta {
        int sid[U];  // neighbor ids of the group's U edges (same addresses in every lane of the group: one request each)
        float w;     // this lane's edge (lane < U): its value (GCN)
    };
    auto meta_load = [&](int base, Meta &m) {
        const int e0 = base + g * U;
#pragma unroll
        for (int u = 0; u < U; ++u) m.sid[u] = a.idx[e0 + u < d.y ? e0 + u : d.y - 1];
        if (!IS_GAT) m.w = a.val ? a.val[e0 + mlane < d.y ? e0 + mlane : d.y - 1] : 1.0f;
    };
    auto issue = [&](const Meta &m, Pack<VEC> (&xv)[U], float &wv) {
        if (IS_GAT) {  // source term of this lane's edge; exp() once it has landed
            int sl = m.sid[0];
#pragma unroll
            for (int u = 1; u < U; ++u) sl = mlane == u ? m.sid[u] : sl;
            wv = a.att[((size_t)sl * a.heads + head) * 2 + 1];
        } else {
            wv = m.w;
        }
        if (col_ok) {
#pragma unroll
            for (int u = 0; u < U; ++u) xv[u] = load_pack<VEC>(xcol + (size_t)m.sid[u] * F);
        }
    };
    // registers -> LDS: quads of 4 consecutive edges per column
    auto stage_round = [&](const Pack<VEC> (&xv)[U], float wv, float *stage, float *wst) {
        if (col_ok) {
#pragma unroll
            for (int hq = 0; hq < U / 4; ++hq) {
                const int kq = (g * U + 4 * hq) >> 2;
                const int swz = (kq >> 1) & 3;
#pragma unroll
                for (int j = 0; j < VEC; ++j)
                    *reinterpret_cast<float4 *>(&stage[(kq * 32 + ((lane * VEC + j) ^ swz)) * 4]) =
                        make_float4(xv[4 * hq].v[j], xv[4 * hq + 1].v[j], xv[4 * hq + 2].v[j], xv[4 * hq + 3].v[j]);
            }
        }
        if (lane < U) wst[g * U + lane] = IS_GAT ? edge_weight(a_dst, wv, a.slope) : wv;
    };
    Pack<VEC> xa[U], xb[U];
    float wa = 0.0f, wb = 0.0f;
    Meta ma, mb;  // metadata of the next issue of set A / set B
    meta_load(d.x, ma);
    meta_load(d.x + RE, mb);
    issue(ma, xa, wa);
    meta_load(d.x + 2 * RE, ma);
    issue(mb, xb, wb);
    meta_load(d.x + 3 * RE, mb);
    for (int r = 0; r < nrounds; r += 2) {
        const int base = d.x + r * RE;
        stage_round(xa, wa, stage0, wst0);
        if (r + 2 < nrounds) {
            issue(ma, xa, wa);                 // round r+2
            meta_load(base + 4 * RE, ma);      // round r+4
        }
        __syncthreads();  // round r is staged (and the consumer is done with round r-1's buffer, which r+1 overwrites)
        if (r + 1 >= nrounds) break;
        stage_round(xb, wb, stage1, wst1);
        if (r + 3 < nrounds) {
            issue(mb, xb, wb);                 // round r+3
            meta_load(base + 5 * RE, mb);      // round r+5
        }
        __syncthreads();
    }
}

struct CombineArgs {
    const int *mrow_id, *mrow_ptr, *row_ptr;
    const int *big_rows;  // indices into mrow_* of the rows with more than kCombineBatch partials
    int n_big, nblocks_small;
    int accumulate;  // 1: y += (sum of partials)
    int relu = 0;    // 1: y = max(result, 0) (GCN)
    // run_with_nn: nn_out[row, :] = (finished row) . nn_weight for the rows finished here (ntiles == 1)
    const float *nn_weight;
    float *nn_out;
    int nn_cols;
    const float *partial;
    const float *partial_den;  // GAT only
    float *y;
    int n_mrows, feat, ntiles, heads, dhead, mean;
};

// Adds the partial rows of every split row in ascending slot order (deterministic counterpart of
// the reference's atomicAdd, aggr_gcn.h:112) and applies mean / softmax normalisation.
static constexpr int kCombineBatch = 16;   // partial rows a lane group keeps in flight
static constexpr int kCombineStage = 128;  // partial rows a workgroup stages in LDS per round (big rows)

template <int VEC, int GROUP, bool IS_MAX, bool IS_GAT>
__global__ __launch_bounds__(kBlock) void k_combine(const CombineArgs a)
{
    constexpr int ITEMS = kBlock / GROUP;
    const int F = a.feat;
    __shared__ float stage[kCombineStage * GROUP * VEC];
    __shared__ float stage_den[IS_GAT ? kCombineStage * 64 : 1];
    const bool nn = !IS_GAT && a.nn_weight != nullptr;
    if ((int)blockIdx.x >= a.nblocks_small) {
        // ---- big rows (hubs: hundreds of partials): one workgroup per (row, column tile).  All lane
        // groups fetch partial rows in parallel into LDS (kCombineStage rows per round, kCombineBatch
        // loads in flight per group), then each column is summed from LDS in ascending slot order.
        const int bb = (int)blockIdx.x - a.nblocks_small;
        const int tile = bb % a.ntiles;
        const int m = a.big_rows[bb / a.ntiles];
        const int s0 = a.mrow_ptr[m], s1 = a.mrow_ptr[m + 1];
        const int row = a.mrow_id[m];
        const int grp = (int)threadIdx.x / GROUP, lane = threadIdx.x & (GROUP - 1);
        const int col0 = tile * GROUP * VEC;
        const int col = col0 + lane * VEC;
        constexpr int W = GROUP * VEC;                // columns of this tile
        const int c = (int)threadIdx.x;               // summing thread <-> column c of the tile
        const bool sum_ok = c < W && col0 + c < F;
        const int hc = IS_GAT ? (col0 + c) / a.dhead : 0;
        float acc = IS_MAX ? -INFINITY : 0.0f, den = 0.0f;
        for (int sb = s0; sb < s1; sb += kCombineStage) {
            const int nst = s1 - sb < kCombineStage ? s1 - sb : kCombineStage;
            for (int p0 = grp * kCombineBatch; p0 < nst; p0 += ITEMS * kCombineBatch) {
                Pack<VEC> p[kCombineBatch];
#pragma unroll
                for (int u = 0; u < kCombineBatch; ++u)
                    if (p0 + u < nst && col < F) p[u] = load_pack<VEC>(a.partial + (size_t)(sb + p0 + u) * F + col);
#pragma unroll
                for (int u = 0; u < kCombineBatch; ++u)
                    if (p0 + u < nst && col < F) store_pack<VEC>(&stage[(p0 + u) * W + lane * VEC], p[u].v);
            }
            if (IS_GAT)
                for (int i = threadIdx.x; i < nst * a.heads; i += kBlock)
                    stage_den[i] = a.partial_den[(size_t)sb * a.heads + i];
            __syncthreads();
            if (sum_ok) {
                // LDS reads issued 16 at a time; the adds stay in ascending order
                for (int p0 = 0; p0 < nst; p0 += 16) {
                    float v[16], dv[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u)
                        if (p0 + u < nst) {
                            v[u] = stage[(p0 + u) * W + c];
                            if (IS_GAT) dv[u] = stage_den[(p0 + u) * a.heads + hc];
                        }
#pragma unroll
                    for (int u = 0; u < 16; ++u)
                        if (p0 + u < nst) {
                            if (IS_MAX) acc = v[u] > acc ? v[u] : acc; else acc += v[u];
                            if (IS_GAT) den += dv[u];
                        }
                }
            }
            __syncthreads();
        }
        if (sum_ok) {
            if (IS_GAT) {
                if (den != 0.0f) acc = acc / den;
            } else if (a.mean) {
                acc = acc / (float)(a.row_ptr[row + 1] - a.row_ptr[row]);
            }
            if (a.accumulate) acc = a.y[(size_t)row * F + col0 + c] + acc;
            if (!IS_GAT && a.relu) acc = acc > 0.0f ? acc : 0.0f;
            a.y[(size_t)row * F + col0 + c] = acc;
            if (nn) stage[c] = acc;  // the staging rounds are over
        }
        if (!nn) return;
        __syncthreads();
        row_times_weight(stage, F, a.nn_weight, a.nn_cols, a.nn_out + (size_t)row * a.nn_cols, (int)threadIdx.x, kBlock);
        return;
    }
    const int tile = blockIdx.x % a.ntiles;
    const int grp = (int)threadIdx.x / GROUP;
    const int m = (blockIdx.x / a.ntiles) * ITEMS + grp;
    const int lane = threadIdx.x & (GROUP - 1);
    const int col = (tile * GROUP + lane) * VEC;
    bool here = m < a.n_mrows;  // this lane group finishes row m (lane-group uniform)
    int s0 = 0, s1 = 0;
    if (here) {
        s0 = a.mrow_ptr[m];
        s1 = a.mrow_ptr[m + 1];
        if (a.n_big > 0 && s1 - s0 > kCombineBatch) here = false;  // handled by the workgroup-per-row path
    }
    const bool active = here && col < a.feat;
    if (!nn && !active) return;
    const int row = here ? a.mrow_id[m] : 0;
    if (active) {
        float acc[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = IS_MAX ? -INFINITY : 0.0f;
        float den = 0.0f;
        const int h = IS_GAT ? col / a.dhead : 0;
        // the adds stay in ascending slot order; only the loads are batched (a hub row of a power-law
        // graph has hundreds of partials -- one dependent load per iteration made this kernel slower
        // than the aggregation itself)
        constexpr int CU = kCombineBatch;
        for (int sb = s0; sb < s1; sb += CU) {
            Pack<VEC> p[CU];
            float pd[CU];
#pragma unroll
            for (int u = 0; u < CU; ++u)
                if (sb + u < s1) {
                    p[u] = load_pack<VEC>(a.partial + (size_t)(sb + u) * F + col);
                    if (IS_GAT) pd[u] = a.partial_den[(size_t)(sb + u) * a.heads + h];
                }
#pragma unroll
            for (int u = 0; u < CU; ++u)
                if (sb + u < s1) {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) {
                        if (IS_MAX)
                            acc[k] = p[u].v[k] > acc[k] ? p[u].v[k] : acc[k];
                        else
                            acc[k] += p[u].v[k];
                    }
                    if (IS_GAT) den += pd[u];
                }
        }
        if (IS_GAT) {
            if (den != 0.0f) {  // scaleArray, aggr_gat.h:207-213
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[k] = acc[k] / den;
            }
        } else if (a.mean) {
            const float d = (float)(a.row_ptr[row + 1] - a.row_ptr[row]);
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] = acc[k] / d;
        }
        if (a.accumulate) {
            const Pack<VEC> old = load_pack<VEC>(a.y + (size_t)row * F + col);
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] = old.v[k] + acc[k];
        }
        if (!IS_GAT && a.relu) relu_pack<VEC>(acc);
        store_pack<VEC>(a.y + (size_t)row * F + col, acc);
        if (nn) store_pack<VEC>(&stage[grp * GROUP * VEC + col], acc);  // ntiles == 1: col = lane * VEC
    }
    if (!nn) return;
    __syncthreads();
    if (here) row_times_weight(&stage[grp * GROUP * VEC], F, a.nn_weight, a.nn_cols, a.nn_out + (size_t)row * a.nn_cols, lane, GROUP);
}

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

__device__ __forceinline__ float edge_weight(float a_dst, float a_src, float slope)
{
    // reference aggr_gat.h:138-143: exp(max(s, s*slope)), no max-subtraction
    const float sc = a_dst + a_src;
    const float l = sc * slope;
    return expf(sc > l ? sc : l);
}

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
template <int VEC, int GROUP>
__device__ __forceinline__ void chain_edges_gat(float (&acc)[VEC], float &den, int beg, int end, int lane, bool col_ok,
                                                const int *__restrict__ idx, const float *__restrict__ att_src, int H,
                                                float a_dst, float slope, const float *__restrict__ xcol, int F,
                                                float *newval, int h, bool head_leader)
{
    int my_s = 0;
    if (beg + lane < end) my_s = idx[beg + lane];
    for (int cb = beg; cb < end; cb += GROUP) {
        int nx_s = 0;
        if (cb + GROUP + lane < end) nx_s = idx[cb + GROUP + lane];
        const int n = end - cb < GROUP ? end - cb : GROUP;
        for (int j = 0; j < n; j += kUnroll) {
            int s[kUnroll];
            float as[kUnroll];
            Pack<VEC> xv[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) s[u] = __shfl(my_s, j + u, GROUP);
#pragma unroll
            for (int u = 0; u < kUnroll; ++u)
                if (j + u < n && col_ok) {
                    as[u] = att_src[(size_t)s[u] * H * 2];
                    xv[u] = load_pack<VEC>(xcol + (size_t)s[u] * F);
                }
#pragma unroll
            for (int u = 0; u < kUnroll; ++u)
                if (j + u < n && col_ok) {
                    const float w = edge_weight(a_dst, as[u], slope);
                    if (newval && head_leader) newval[(size_t)(cb + j + u) * H + h] = w;
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
template <int VEC, int GROUP>
__device__ __forceinline__ void chain_edges_gat1(float (&acc)[VEC], float &den, int beg, int end, int lane, bool col_ok,
                                                 const int *__restrict__ idx, const float *__restrict__ att_src, float a_dst,
                                                 float slope, const float *__restrict__ xcol, int F, float *newval,
                                                 bool first_tile)
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
        for (int j = 0; j < n; j += kUnroll) {
            int s[kUnroll];
            float w[kUnroll];
            Pack<VEC> xv[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) s[u] = __shfl(s0, j + u, GROUP);
#pragma unroll
            for (int u = 0; u < kUnroll; ++u)
                if (j + u < n && col_ok) xv[u] = load_pack<VEC>(xcol + (size_t)s[u] * F);
            if (j == 0) {  // this lane's edge of the window
                my_w = lane < n ? edge_weight(a_dst, a0, slope) : 0.0f;
                if (newval && first_tile && lane < n) newval[cb + lane] = my_w;
            }
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) w[u] = __shfl(my_w, j + u, GROUP);
#pragma unroll
            for (int u = 0; u < kUnroll; ++u)
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
    XcdRanges xr;
};

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
    __builtin_amdgcn_s_waitcnt(0);  // the write-through stores have reached the device coherence point
    __shared__ int s_hub;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int m = a.slot_hub[slot];
        const int nseg = a.mrow_ptr[m + 1] - a.mrow_ptr[m];
        int *cnt = a.hub_count + (size_t)m * a.hub_count_stride + tile;
        const int old = atomicAdd(cnt, 1);
        if (old == nseg - 1) atomicExch(cnt, 0);
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
        if (den != 0.0f) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] = acc[k] / den;
        }
        store_pack<VEC>(a.y + (size_t)row * F + col, acc);
    }
}

// (forcing 6 waves/SIMD -- 80 VGPRs, 5-9 spilled -- changes nothing: 100.8 vs 101.5 us on fig10a, 13.8 vs 13.6 ms on config G)
template <int VEC, int GROUP, bool SINGLE>
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
        const int b = logical_block((int)blockIdx.x - nb1, a.nblocks0, a.ntiles, a.remap, a.xr);
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
    const float *__restrict__ xcol = a.x + col;
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
                chain_edges_gat1<VEC, GROUP>(acc, den, cb, ce, lane, col_ok, a.idx, att_src, a_dst, a.slope, xcol, F, a.newval,
                                             tile == 0);
            else
                chain_edges_gat<VEC, GROUP>(acc, den, cb, ce, lane, col_ok, a.idx, att_src, H, a_dst, a.slope, xcol, F, a.newval,
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
            if (den != 0.0f) {  // scaleArray, aggr_gat.h:207-213
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[k] = acc[k] / den;
            }
            store_pack<VEC>(a.y + (size_t)row_or_dest * F + col, acc);
        } else {
            store_pack<VEC>(a.partial + (size_t)(~row_or_dest) * F + col, acc);
            if (head_leader) a.partial_den[(size_t)(~row_or_dest) * H + h] = den;
        }
        return;
    }
    const int row = d.z;
    float acc[VEC] = {};
    float den = 0.0f;
    if (d.x < d.y) {
        const float a_dst = a.att[((size_t)row * H + h) * 2];
        if constexpr (SINGLE)
            chain_edges_gat1<VEC, GROUP>(acc, den, d.x, d.y, lane, col_ok, a.idx, att_src, a_dst, a.slope, xcol, F, a.newval,
                                         tile == 0);
        else
            chain_edges_gat<VEC, GROUP>(acc, den, d.x, d.y, lane, col_ok, a.idx, att_src, H, a_dst, a.slope, xcol, F, a.newval, h,
                                        head_leader);
    }
    if (!col_ok) return;
    if (d.x < d.y && (den != 0.0f || a.rows_semantics)) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = acc[k] / den;
    }
    store_pack<VEC>(a.y + (size_t)row * F + col, acc);
}

// --------------------------------------------------------------------- geometry + dispatch
struct Geometry {
    int vec, group, ntiles;
};

static Geometry pick_geometry(int F, const void *p0, const void *p1, const void *p2, int dhead)
{
    auto aligned = [](const void *p, size_t a) { return p == nullptr || ((uintptr_t)p % a) == 0; };
    int vec = 1;
    if (F % 4 == 0 && dhead % 4 == 0 && aligned(p0, 16) && aligned(p1, 16) && aligned(p2, 16))
        vec = 4;
    else if (F % 2 == 0 && dhead % 2 == 0 && aligned(p0, 8) && aligned(p1, 8) && aligned(p2, 8))
        vec = 2;
    const int lanes = (F + vec - 1) / vec;
    int group = 8;
    while (group < 64 && group < lanes) group <<= 1;
    const int ntiles = (lanes + group - 1) / group;
    return {vec, group, ntiles};
}

#define DISPATCH_GEOM(g, KERNEL_CALL)                                            \
    switch ((g).vec * 100 + (g).group) {                                         \
        case 108: { constexpr int VEC = 1, GROUP = 8;  KERNEL_CALL; } break;     \
        case 116: { constexpr int VEC = 1, GROUP = 16; KERNEL_CALL; } break;     \
        case 132: { constexpr int VEC = 1, GROUP = 32; KERNEL_CALL; } break;     \
        case 164: { constexpr int VEC = 1, GROUP = 64; KERNEL_CALL; } break;     \
        case 208: { constexpr int VEC = 2, GROUP = 8;  KERNEL_CALL; } break;     \
        case 216: { constexpr int VEC = 2, GROUP = 16; KERNEL_CALL; } break;     \
        case 232: { constexpr int VEC = 2, GROUP = 32; KERNEL_CALL; } break;     \
        case 264: { constexpr int VEC = 2, GROUP = 64; KERNEL_CALL; } break;     \
        case 408: { constexpr int VEC = 4, GROUP = 8;  KERNEL_CALL; } break;     \
        case 416: { constexpr int VEC = 4, GROUP = 16; KERNEL_CALL; } break;     \
        case 432: { constexpr int VEC = 4, GROUP = 32; KERNEL_CALL; } break;     \
        case 464: { constexpr int VEC = 4, GROUP = 64; KERNEL_CALL; } break;     \
        default: return fail(GNNAGG_ERR_ARG, "unsupported lane geometry");       \
    }

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// cost_prefix[i] = total cost of items [0,i) (host array, n_items+1 entries).  Cuts the item blocks
// into 8 contiguous ranges of about equal cost; returns the longest range (in blocks).
static int fill_xcd_ranges(const long *cost_prefix, int n_items, int items_per_block, int item_blocks, XcdRanges &xr)
{
    const long total = cost_prefix[n_items];
    int start = 0, longest = 0;
    for (int x = 0; x < 8; ++x) {
        int stop;
        if (x == 7) {
            stop = item_blocks;
        } else {
            const long want = total * (x + 1) / 8;
            // first block boundary whose prefix cost reaches `want`
            int lo = start, hi = item_blocks;
            while (lo < hi) {
                const int mid = (lo + hi) / 2;
                const long c = cost_prefix[std::min((long)mid * items_per_block, (long)n_items)];
                if (c < want) lo = mid + 1; else hi = mid;
            }
            stop = lo;
        }
        xr.first[x] = start;
        xr.count[x] = stop - start;
        longest = std::max(longest, stop - start);
        start = stop;
    }
    return longest;
}

static int launch_combine_gcn(const GcnLaunch &L, const Geometry &g, bool is_max, hipStream_t stream,
                              const float *nn_weight = nullptr, float *nn_out = nullptr, int nn_cols = 0)
{
    if (L.wl.n_mrows > 0) {
        CombineArgs c;
        c.mrow_id = L.wl.mrow_id; c.mrow_ptr = L.wl.mrow_ptr; c.row_ptr = L.row_ptr; c.partial = L.partial;
        c.partial_den = nullptr; c.y = L.y; c.n_mrows = L.wl.n_mrows; c.feat = L.feat; c.ntiles = g.ntiles;
        c.heads = 1; c.dhead = L.feat; c.mean = L.reduce == GNNAGG_REDUCE_MEAN;
        c.accumulate = L.accumulate; c.relu = L.relu;
        c.nn_weight = nn_weight; c.nn_out = nn_out; c.nn_cols = nn_cols;
        c.big_rows = L.wl.big_rows; c.n_big = L.wl.n_big;
        c.nblocks_small = ceil_div(c.n_mrows, kBlock / g.group) * g.ntiles;
        const int nb = c.nblocks_small + c.n_big * g.ntiles;
#define CALL_COMB                                                                                           \
        if (is_max) hipLaunchKernelGGL((k_combine<VEC, GROUP, true, false>), dim3(nb), dim3(kBlock), 0, stream, c);  \
        else        hipLaunchKernelGGL((k_combine<VEC, GROUP, false, false>), dim3(nb), dim3(kBlock), 0, stream, c);
        DISPATCH_GEOM(g, CALL_COMB)
#undef CALL_COMB
        HIP_TRY(hipGetLastError());
    }
    return GNNAGG_OK;
}

int launch_dense_nn(const float *A, const float *B, float *C, int M, int N, int K, void *stream_v);

static bool nn_fusion_enabled()
{
    static const int on = getenv("GNNAGG_FUSE_NN") ? atoi(getenv("GNNAGG_FUSE_NN")) : 1;
    return on != 0;
}

int launch_dense_rows(const int *rows, int n_rows, const float *Y, const float *W, float *out, int K, int N, void *stream_v)
{
    hipStream_t stream = (hipStream_t)stream_v;
    if (n_rows <= 0 || N <= 0) return GNNAGG_OK;
    if ((size_t)K * sizeof(float) > 60 * 1024) return fail(GNNAGG_ERR_ARG, "dense_rows: feature length too large");
    hipLaunchKernelGGL(k_dense_rows, dim3(n_rows), dim3(kBlock), (size_t)K * sizeof(float), stream, rows, Y, W, out, K, N);
    HIP_TRY(hipGetLastError());
    return GNNAGG_OK;
}

int launch_gcn_plan(const GcnPlanLaunch &L, void *stream_v)
{
    hipStream_t stream = (hipStream_t)stream_v;
    if (L.feat <= 0) return fail(GNNAGG_ERR_ARG, "feature length must be >= 1");
    const Geometry g = pick_geometry(L.feat, L.x, L.y, L.partial, L.feat);
    const bool is_max = L.reduce == GNNAGG_REDUCE_MAX;
    PlanArgs a;
    a.t0 = reinterpret_cast<const int4 *>(L.t0); a.t1 = reinterpret_cast<const int4 *>(L.t1);
    a.idx = L.idx; a.val = L.val; a.x = L.x; a.y = L.y; a.partial = L.partial;
    a.n0 = L.n0; a.n1 = L.n1; a.feat = L.feat; a.ntiles = g.ntiles; a.chunk = L.chunk;
    a.mean = L.reduce == GNNAGG_REDUCE_MEAN; a.remap = L.xcd_remap; a.accumulate = L.accumulate; a.relu = L.relu;
    a.slot_hub = L.slot_hub; a.mrow_ptr = L.hubs.mrow_ptr; a.mrow_id = L.hubs.mrow_id; a.row_ptr = L.row_ptr;
    a.hub_count = L.hub_count; a.hub_count_stride = L.hub_count_stride; a.partial_bytes = 0;
    {
        const size_t pbytes = (size_t)L.hubs.n_slots * L.feat * sizeof(float);
        if (L.hubs.n_mrows == 0 || pbytes >= 0x7fffffffULL || g.ntiles > L.hub_count_stride) a.hub_count = nullptr;
        else a.partial_bytes = (unsigned)pbytes;
    }
    const bool hubs_in_kernel = a.hub_count != nullptr;
    {
        static const int wt_env = getenv("GNNAGG_WT_STORES") ? atoi(getenv("GNNAGG_WT_STORES")) : 1;
        const size_t ybytes = (size_t)L.num_rows * L.feat * sizeof(float);
        a.wt = (wt_env && !L.accumulate && ybytes < 0x7fffffffULL) ? 1 : 0;
        a.ybytes = (unsigned)ybytes;
    }
    // dense combine fused as the epilogue when one lane group spans the row and the [32][K] tile fits LDS
    const bool want_nn = L.nn_weight != nullptr;
    // (8-lane groups, F <= 32: the GEMM is ~11 us on the arxiv-shaped input and the epilogue costs as much -- not fused)
    const bool fuse_nn = want_nn && g.ntiles == 1 && g.group >= 16 && !L.accumulate && !L.relu && nn_fusion_enabled();
    const int blk = block_for(g.group);
    const int gpb = fuse_nn ? std::max(kNnRows, blk / g.group) : blk / g.group;
    const int item_blocks = ceil_div(a.n0, gpb);
    a.nblocks0 = item_blocks * g.ntiles;
    if (a.remap && a.nblocks0 < 64) a.remap = 0;
    int grid0 = a.nblocks0;
    if (a.remap == 2) {
        if (!L.t0_cost_prefix) a.remap = 1;
        else grid0 = 8 * fill_xcd_ranges(L.t0_cost_prefix, a.n0, gpb, item_blocks, a.xr) * g.ntiles;
    }
    const int grid = a.n1 * g.ntiles + grid0;
    if (fuse_nn) {
        NnArgs w;
        w.weight = L.nn_weight; w.out = L.nn_out; w.n_out = L.nn_cols;
        if (grid > 0) {
#define CALL_PLAN_NN                                                                                                 \
            if (is_max) hipLaunchKernelGGL((k_gcn_plan_nn<VEC, GROUP, true>), dim3(grid), dim3(blk), 0, stream, a, w);     \
            else        hipLaunchKernelGGL((k_gcn_plan_nn<VEC, GROUP, false>), dim3(grid), dim3(blk), 0, stream, a, w);
            DISPATCH_GEOM(g, CALL_PLAN_NN)
#undef CALL_PLAN_NN
            HIP_TRY(hipGetLastError());
        }
        GcnLaunch C;
        C.wl = L.hubs; C.row_ptr = L.row_ptr; C.y = L.y; C.partial = L.partial; C.feat = L.feat; C.reduce = L.reduce;
        if (hubs_in_kernel) return GNNAGG_OK;
        return launch_combine_gcn(C, g, is_max, stream, L.nn_weight, L.nn_out, L.nn_cols);  // hubs: product in the combine
    }
    if (grid > 0) {
#define CALL_PLAN                                                                                            \
        if (is_max) hipLaunchKernelGGL((k_gcn_plan<VEC, GROUP, true>), dim3(grid), dim3(blk), 0, stream, a);        \
        else        hipLaunchKernelGGL((k_gcn_plan<VEC, GROUP, false>), dim3(grid), dim3(blk), 0, stream, a);
        DISPATCH_GEOM(g, CALL_PLAN)
#undef CALL_PLAN
        HIP_TRY(hipGetLastError());
    }
    GcnLaunch C;
    C.wl = L.hubs; C.row_ptr = L.row_ptr; C.y = L.y; C.partial = L.partial; C.feat = L.feat; C.reduce = L.reduce;
    C.accumulate = L.accumulate; C.relu = L.relu;
    const int rc = hubs_in_kernel ? GNNAGG_OK : launch_combine_gcn(C, g, is_max, stream);
    if (rc || !want_nn) return rc;
    return launch_dense_nn(L.y, L.nn_weight, L.nn_out, L.num_rows, L.nn_cols, L.feat, stream);
}

int launch_gcn_rows_long(const GcnRowsLongLaunch &L, void *stream_v)
{
    hipStream_t stream = (hipStream_t)stream_v;
    if (L.n1 <= 0) return GNNAGG_OK;
    auto aligned = [](const void *p, size_t al) { return ((uintptr_t)p % al) == 0; };
    int vec = 1;
    if (L.feat % 4 == 0 && aligned(L.x, 16)) vec = 4;
    else if (L.feat % 2 == 0 && aligned(L.x, 8)) vec = 2;
    RowsLongArgs a;
    a.r1 = reinterpret_cast<const int4 *>(L.r1); a.idx = L.idx; a.val = L.val; a.x = L.x; a.y = L.y;
    a.n1 = L.n1; a.feat = L.feat; a.ntiles32 = ceil_div(L.feat, 32); a.mean = L.reduce == GNNAGG_REDUCE_MEAN; a.relu = L.relu;
    a.att = L.att; a.heads = L.heads; a.dhead = L.heads > 0 ? L.feat / L.heads : L.feat; a.slope = L.slope;
    const bool is_max = L.reduce == GNNAGG_REDUCE_MAX;
    const bool is_gat = L.att != nullptr;
    if (is_gat && (a.dhead % 32) != 0) return fail(GNNAGG_ERR_ARG, "long-row GAT kernel needs head width % 32 == 0");
    const int grid = a.n1 * a.ntiles32;
#define LAUNCH_LONG(V)                                                                                               \
    {                                                                                                                \
        const size_t lds = (size_t)long_round_edges<V>() * (2 * 32 + 2) * sizeof(float);                             \
        if (is_gat)      LAUNCH_LONG_K((k_gcn_rows_long<V, false, true>))                                            \
        else if (is_max) LAUNCH_LONG_K((k_gcn_rows_long<V, true, false>))                                            \
        else             LAUNCH_LONG_K((k_gcn_rows_long<V, false, false>))                                           \
    }
#define LAUNCH_LONG_K(K)                                                                                             \
    {                                                                                                                \
        static bool big_lds_ok = false; /* > 64 KB of dynamic LDS needs the attribute, once per instantiation */    \
        if (!big_lds_ok) {                                                                                           \
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&K), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            big_lds_ok = true;                                                                                       \
        }                                                                                                            \
        hipLaunchKernelGGL(K, dim3(grid), dim3(kLongBlock), lds, stream, a);                                         \
    }
    if (vec == 4) LAUNCH_LONG(4) else if (vec == 2) LAUNCH_LONG(2) else LAUNCH_LONG(1)
#undef LAUNCH_LONG_K
#undef LAUNCH_LONG
    HIP_TRY(hipGetLastError());
    return GNNAGG_OK;
}

int launch_gcn(const GcnLaunch &L, void *stream_v)
{
    hipStream_t stream = (hipStream_t)stream_v;
    if (L.feat <= 0) return fail(GNNAGG_ERR_ARG, "feature length must be >= 1");
    const bool list = L.wl.target != nullptr || L.wl.slot != nullptr || L.wl.n_empty > 0;
    const Geometry g = pick_geometry(L.feat, L.x, L.y, L.partial, L.feat);
    const bool is_max = L.reduce == GNNAGG_REDUCE_MAX;
    GcnArgs a;
    a.ptr = L.wl.ptr; a.target = L.wl.target; a.slot = L.wl.slot; a.empty_rows = L.wl.empty_rows;
    a.row_ptr = L.row_ptr; a.idx = L.idx; a.val = L.val; a.x = L.x; a.y = L.y; a.partial = L.partial;
    a.n_items = L.wl.n_items; a.n_total = L.wl.n_items + L.wl.n_empty; a.feat = L.feat; a.ntiles = g.ntiles;
    a.mean = L.reduce == GNNAGG_REDUCE_MEAN; a.remap = L.xcd_remap; a.relu = L.relu;
    a.timer = reinterpret_cast<unsigned long long *>(L.timer);
    if (L.timer || L.timer_blocks_out) a.remap = 0;  // natural block order for the load-balance study
    if (L.timer_blocks_out) {
        *L.timer_blocks_out = a.n_total > 0 ? ceil_div(a.n_total, block_for(g.group) / g.group) * g.ntiles : 0;
        if (!L.timer) return GNNAGG_OK;  // size query only
    }
    if (a.n_total > 0) {
        const int blk = block_for(g.group);
        const int items_per_block = blk / g.group;
        const int item_blocks = ceil_div(a.n_total, items_per_block);
        a.nblocks = item_blocks * g.ntiles;
        if (a.remap && a.nblocks < 64) a.remap = 0;
        int grid = a.nblocks;
        if (a.remap == 2) {
            if (!L.xcd_item_cost_prefix) {
                a.remap = 1;
            } else {
                grid = 8 * fill_xcd_ranges(L.xcd_item_cost_prefix, a.n_total, items_per_block, item_blocks, a.xr) * g.ntiles;
            }
        }
        if (a.timer) hipLaunchKernelGGL(k_timer_init, dim3(ceil_div(grid, 256)), dim3(256), 0, stream, a.timer, grid);
#define LAUNCH_GCN(MAXF, LISTF) hipLaunchKernelGGL((k_gcn_items<VEC, GROUP, MAXF, LISTF>), dim3(grid), dim3(blk), 0, stream, a)
#define CALL_GCN                                                                                  \
        if (list) { if (is_max) LAUNCH_GCN(true, true); else LAUNCH_GCN(false, true); }           \
        else      { if (is_max) LAUNCH_GCN(true, false); else LAUNCH_GCN(false, false); }
        DISPATCH_GEOM(g, CALL_GCN)
#undef CALL_GCN
#undef LAUNCH_GCN
        HIP_TRY(hipGetLastError());
    }
    { int rc = launch_combine_gcn(L, g, is_max, stream); if (rc) return rc; }
    return GNNAGG_OK;
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
        c.mrow_id = L.wl.mrow_id; c.mrow_ptr = L.wl.mrow_ptr; c.row_ptr = nullptr; c.partial = L.partial;
        c.partial_den = L.partial_den; c.y = L.y; c.n_mrows = L.wl.n_mrows; c.feat = L.feat; c.ntiles = g.ntiles;
        c.heads = L.heads; c.dhead = dhead; c.mean = 0; c.accumulate = 0;
        c.big_rows = L.wl.big_rows; c.n_big = L.heads <= 64 ? L.wl.n_big : 0;
        c.nblocks_small = ceil_div(c.n_mrows, kBlock / g.group) * g.ntiles;
        const int nb = c.nblocks_small + c.n_big * g.ntiles;
#define CALL_COMB hipLaunchKernelGGL((k_combine<VEC, GROUP, false, true>), dim3(nb), dim3(kBlock), 0, stream, c);
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
    const Geometry g = pick_geometry(L.feat, L.x, L.y, L.partial, dhead);
    GatPlanArgs a;
    a.t0 = reinterpret_cast<const int4 *>(L.t0); a.t1 = reinterpret_cast<const int4 *>(L.t1);
    a.idx = L.idx; a.att = L.att; a.x = L.x; a.y = L.y; a.partial = L.partial; a.partial_den = L.partial_den;
    a.newval = L.newval; a.n0 = L.n0; a.n1 = L.n1; a.feat = L.feat; a.ntiles = g.ntiles; a.chunk = L.chunk;
    a.heads = L.heads; a.dhead = dhead; a.remap = L.xcd_remap; a.slope = L.slope; a.rows_semantics = L.rows_semantics;
    a.slot_hub = L.slot_hub; a.mrow_ptr = L.hubs.mrow_ptr; a.mrow_id = L.hubs.mrow_id;
    a.hub_count = L.hub_count; a.hub_count_stride = L.hub_count_stride; a.partial_bytes = a.partial_den_bytes = 0;
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
    if (a.remap && a.nblocks0 < 64) a.remap = 0;
    int grid0 = a.nblocks0;
    if (a.remap == 2) {
        if (!L.t0_cost_prefix) a.remap = 1;
        else grid0 = 8 * fill_xcd_ranges(L.t0_cost_prefix, a.n0, gpb, item_blocks, a.xr) * g.ntiles;
    }
    const int grid = a.n1 * g.ntiles + grid0;
    if (grid > 0) {
#define CALL_GP                                                                                              \
        if (a.heads == 1) hipLaunchKernelGGL((k_gat_plan<VEC, GROUP, true>), dim3(grid), dim3(blk), 0, stream, a);   \
        else              hipLaunchKernelGGL((k_gat_plan<VEC, GROUP, false>), dim3(grid), dim3(blk), 0, stream, a);
        DISPATCH_GEOM(g, CALL_GP)
#undef CALL_GP
        HIP_TRY(hipGetLastError());
    }
    if (L.hubs.n_mrows > 0 && !hubs_in_kernel) {
        CombineArgs c;
        c.mrow_id = L.hubs.mrow_id; c.mrow_ptr = L.hubs.mrow_ptr; c.row_ptr = nullptr; c.partial = L.partial;
        c.partial_den = L.partial_den; c.y = L.y; c.n_mrows = L.hubs.n_mrows; c.feat = L.feat; c.ntiles = g.ntiles;
        c.heads = L.heads; c.dhead = dhead; c.mean = 0; c.accumulate = 0;
        c.big_rows = L.hubs.big_rows; c.n_big = L.heads <= 64 ? L.hubs.n_big : 0;
        c.nblocks_small = ceil_div(c.n_mrows, kBlock / g.group) * g.ntiles;
        const int nb = c.nblocks_small + c.n_big * g.ntiles;
#define CALL_COMB hipLaunchKernelGGL((k_combine<VEC, GROUP, false, true>), dim3(nb), dim3(kBlock), 0, stream, c);
        DISPATCH_GEOM(g, CALL_COMB)
#undef CALL_COMB
        HIP_TRY(hipGetLastError());
    }
    return GNNAGG_OK;
}

// ------------------------------------------------------------------------- CSR -> edge list
// A lane group strides over the edges of one row; group size follows the average degree so short
// rows do not idle a whole wavefront.
static int edge_group(int avg_deg)
{
    int g = 8;
    while (g < 64 && g < avg_deg) g <<= 1;
    return g;
}

template <int GROUP>
__device__ __forceinline__ float group_sum(float v)
{
#pragma unroll
    for (int m = GROUP / 2; m > 0; m >>= 1) v += __shfl_xor(v, m, GROUP);
    return v;
}

// reference convertCSRToEdgelist, aggregator.h:11-23 ((src,dst) written as one 8-byte store)
template <int GROUP>
__global__ __launch_bounds__(kBlock) void k_csr2edgelist(const int *__restrict__ ptr, const int *__restrict__ idx,
                                                        int2 *__restrict__ edgelist, int V)
{
    const int row = blockIdx.x * (kBlock / GROUP) + threadIdx.x / GROUP;
    const int lane = threadIdx.x & (GROUP - 1);
    if (row >= V) return;
    for (int e = ptr[row] + lane; e < ptr[row + 1]; e += GROUP) edgelist[e] = make_int2(idx[e], row);
}

#define DISPATCH_EDGE_GROUP(G, CALL)                      \
    switch (G) {                                          \
        case 8:  { constexpr int GROUP = 8;  CALL; } break;  \
        case 16: { constexpr int GROUP = 16; CALL; } break;  \
        case 32: { constexpr int GROUP = 32; CALL; } break;  \
        default: { constexpr int GROUP = 64; CALL; } break;  \
    }

int launch_csr2edgelist(const int *ptr, const int *idx, int *edgelist, int V, int avg_deg, void *stream_v)
{
    hipStream_t stream = (hipStream_t)stream_v;
    if (V <= 0) return GNNAGG_OK;
    const int G = edge_group(avg_deg);
    const int nb = ceil_div(V, kBlock / G);
    DISPATCH_EDGE_GROUP(G, hipLaunchKernelGGL((k_csr2edgelist<GROUP>), dim3(nb), dim3(kBlock), 0, stream, ptr, idx,
                                              reinterpret_cast<int2 *>(edgelist), V))
    HIP_TRY(hipGetLastError());
    return GNNAGG_OK;
}

// ------------------------------------------------- edge kernels on chunked work items (hub-safe)
// Giving one lane group a whole row (the reference's warp-per-row attGat / u_add_v / add_to_center / each_div,
// aggr_gat.h:5-92) serialises on a 15 k-edge hub row: 860 us for attGat on the arxiv-shaped graph in the first
// version of this file.  These kernels run on the work items of the balanced
// neighbor grouping (<= chunk edges each): pass 1 writes the edge values and per-item sums (straight to
// den[row] when the row has one item, to partial_den[slot] otherwise), an ordered combine finishes the
// split rows, pass 2 normalises.  Lanes walk the flattened (edge, head) pairs of an item, so out[e,h]
// stores are fully coalesced and a lane keeps one head when GROUP % H == 0.
struct EdgeItemArgs {
    const int *ptr_s, *target, *slot, *empty_rows, *idx;
    const float *att;   // [V,H,2]
    const float *in;    // per-edge input (add_to_center) / per-row divisor (div)
    float *out;         // per-edge output [E,H]
    float *den;         // per-row sums [V,H]
    float *partial_den; // [n_slots,H]
    int n_items, n_empty, H;
    float slope;
};

// OP 0: attGat pass 1 (w = exp(leaky(a_dst + a_src)) -> out, item sums)   aggr_gat.h:13-19
// OP 1: add_to_center (item sums of in[e])                                 aggr_gat.h:62-73
template <int GROUP, int OP>
__global__ __launch_bounds__(kBlock) void k_edge_items_sum(const EdgeItemArgs a)
{
    const int item = blockIdx.x * (kBlock / GROUP) + (int)threadIdx.x / GROUP;
    const int lane = threadIdx.x & (GROUP - 1);
    const int H = a.H;
    if (item >= a.n_items + a.n_empty) return;
    if (item >= a.n_items) {  // rows without edges: sum = 0
        const int row = a.empty_rows[item - a.n_items];
        for (int h = lane; h < H; h += GROUP) a.den[(size_t)row * H + h] = 0.0f;
        return;
    }
    const int beg = a.ptr_s[item], end = a.ptr_s[item + 1];
    const int row = a.target[item];
    const int sl = a.slot[item];
    float *dst = sl >= 0 ? a.partial_den + (size_t)sl * H : a.den + (size_t)row * H;
    const int n = (end - beg) * H;
    if (H <= GROUP && (GROUP % H) == 0) {
        const int h = lane % H;  // fixed head per lane: the stride GROUP is a multiple of H
        const float a_dst = OP == 0 ? a.att[((size_t)row * H + h) * 2] : 0.0f;
        float part = 0.0f;
        for (int j = lane; j < n; j += GROUP) {
            const int e = beg + j / H;
            float w;
            if (OP == 0) {
                w = edge_weight(a_dst, a.att[((size_t)a.idx[e] * H + h) * 2 + 1], a.slope);
                a.out[(size_t)beg * H + j] = w;
            } else {
                w = a.in[(size_t)beg * H + j];
            }
            part += w;
        }
        for (int msk = GROUP / 2; msk >= H; msk >>= 1) part += __shfl_xor(part, msk, GROUP);
        if (lane < H) dst[lane] = part;
    } else {
        for (int h = 0; h < H; ++h) {  // odd head counts: one head at a time
            const float a_dst = OP == 0 ? a.att[((size_t)row * H + h) * 2] : 0.0f;
            float part = 0.0f;
            for (int e = beg + lane; e < end; e += GROUP) {
                float w;
                if (OP == 0) {
                    w = edge_weight(a_dst, a.att[((size_t)a.idx[e] * H + h) * 2 + 1], a.slope);
                    a.out[(size_t)e * H + h] = w;
                } else {
                    w = a.in[(size_t)e * H + h];
                }
                part += w;
            }
            part = group_sum<GROUP>(part);
            if (lane == 0) dst[h] = part;
        }
    }
}

// ordered sum of the item sums of split rows
__global__ void k_den_combine(const int *__restrict__ mrow_id, const int *__restrict__ mrow_ptr,
                              const float *__restrict__ partial_den, float *__restrict__ den, int n_mrows, int H)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_mrows * H) return;
    const int m = t / H, h = t % H;
    float s = 0.0f;
    for (int p = mrow_ptr[m]; p < mrow_ptr[m + 1]; ++p) s += partial_den[(size_t)p * H + h];
    den[(size_t)mrow_id[m] * H + h] = s;
}

// OP 0: out[e,h] /= den[row,h]   (attGat pass 2 aggr_gat.h:26-29, each_div aggr_gat.h:84-90)
// OP 1: out[e] = att[row,0] + att[idx[e],1]   (u_add_v aggr_gat.h:44-46)
template <int GROUP, int OP>
__global__ __launch_bounds__(kBlock) void k_edge_items_map(const EdgeItemArgs a)
{
    const int item = blockIdx.x * (kBlock / GROUP) + (int)threadIdx.x / GROUP;
    const int lane = threadIdx.x & (GROUP - 1);
    if (item >= a.n_items) return;
    const int beg = a.ptr_s[item], end = a.ptr_s[item + 1];
    const int row = a.target[item];
    const int H = a.H;
    if (OP == 0) {
        const int n = (end - beg) * H;
        for (int j = lane; j < n; j += GROUP) a.out[(size_t)beg * H + j] /= a.in[(size_t)row * H + j % H];
    } else {
        const float a_dst = a.att[(size_t)row * 2];
        for (int e = beg + lane; e < end; e += GROUP) a.out[e] = a_dst + a.att[(size_t)a.idx[e] * 2 + 1];
    }
}

static int edge_item_group(long avg_pairs)
{
    return avg_pairs <= 8 ? 8 : (avg_pairs <= 32 ? 32 : 64);
}

#define DISPATCH_EIG(G, CALL)                               \
    switch (G) {                                            \
        case 8:  { constexpr int GROUP = 8;  CALL; } break; \
        case 32: { constexpr int GROUP = 32; CALL; } break; \
        default: { constexpr int GROUP = 64; CALL; } break; \
    }

static void fill_edge_args(EdgeItemArgs &a, const EdgeItemLaunch &L)
{
    a.ptr_s = L.wl.ptr; a.target = L.wl.target; a.slot = L.wl.slot; a.empty_rows = L.wl.empty_rows; a.idx = L.idx;
    a.att = L.att; a.in = L.in; a.out = L.out; a.den = L.den; a.partial_den = L.partial_den;
    a.n_items = L.wl.n_items; a.n_empty = L.wl.n_empty; a.H = L.heads; a.slope = L.slope;
}

// sums: op 0 = attGat weights + row sums, op 1 = add_to_center
int launch_edge_items_sum(const EdgeItemLaunch &L, int op, void *stream_v)
{
    hipStream_t stream = (hipStream_t)stream_v;
    EdgeItemArgs a;
    fill_edge_args(a, L);
    const int total = a.n_items + a.n_empty;
    if (total > 0) {
        const int G = edge_item_group((long)L.avg_item_edges * L.heads);
        const int nb = ceil_div(total, kBlock / G);
        if (op == 0) { DISPATCH_EIG(G, hipLaunchKernelGGL((k_edge_items_sum<GROUP, 0>), dim3(nb), dim3(kBlock), 0, stream, a)) }
        else         { DISPATCH_EIG(G, hipLaunchKernelGGL((k_edge_items_sum<GROUP, 1>), dim3(nb), dim3(kBlock), 0, stream, a)) }
        HIP_TRY(hipGetLastError());
    }
    if (L.wl.n_mrows > 0) {
        const int n = L.wl.n_mrows * L.heads;
        hipLaunchKernelGGL(k_den_combine, dim3(ceil_div(n, 256)), dim3(256), 0, stream, L.wl.mrow_id, L.wl.mrow_ptr,
                           L.partial_den, L.den, L.wl.n_mrows, L.heads);
        HIP_TRY(hipGetLastError());
    }
    return GNNAGG_OK;
}

// maps: op 0 = divide by the row value, op 1 = u_add_v
int launch_edge_items_map(const EdgeItemLaunch &L, int op, void *stream_v)
{
    hipStream_t stream = (hipStream_t)stream_v;
    EdgeItemArgs a;
    fill_edge_args(a, L);
    if (a.n_items <= 0) return GNNAGG_OK;
    const int G = edge_item_group((long)L.avg_item_edges * L.heads);
    const int nb = ceil_div(a.n_items, kBlock / G);
    if (op == 0) { DISPATCH_EIG(G, hipLaunchKernelGGL((k_edge_items_map<GROUP, 0>), dim3(nb), dim3(kBlock), 0, stream, a)) }
    else         { DISPATCH_EIG(G, hipLaunchKernelGGL((k_edge_items_map<GROUP, 1>), dim3(nb), dim3(kBlock), 0, stream, a)) }
    HIP_TRY(hipGetLastError());
    return GNNAGG_OK;
}

// ------------------------------------------------------------------------ edge-wise variant
// reference aggr_gcn_edgewise, aggr_gcn.h:291-302 (which covers only 32 columns and reads one edge
// past the end, :296); here one 64-lane wavefront per edge strides over all F columns.
__global__ __launch_bounds__(kBlock) void k_edgewise(const int2 *__restrict__ edgelist, const float *__restrict__ val,
                                                    const float *__restrict__ x, float *__restrict__ y, int E,
                                                    int F)
{
    const int e = blockIdx.x * (kBlock / 64) + threadIdx.x / 64;
    const int lane = threadIdx.x & 63;
    if (e >= E) return;
    const int2 sd = edgelist[e];
    const float w = val ? val[e] : 1.0f;
    for (int c = lane; c < F; c += 64) atomicAdd(&y[(size_t)sd.y * F + c], x[(size_t)sd.x * F + c] * w);
}

int launch_edgewise(const int *edgelist, const float *val, const float *x, float *y, int E, int V, int feat,
                    void *stream_v)
{
    hipStream_t stream = (hipStream_t)stream_v;
    HIP_TRY(hipMemsetAsync(y, 0, (size_t)V * feat * sizeof(float), stream));  // aggr_gcn.h:448
    if (E <= 0) return GNNAGG_OK;
    hipLaunchKernelGGL(k_edgewise, dim3(ceil_div(E, kBlock / 64)), dim3(kBlock), 0, stream,
                       reinterpret_cast<const int2 *>(edgelist), val, x, y, E, feat);
    HIP_TRY(hipGetLastError());
    return GNNAGG_OK;
}

// --------------------------------------------------------------------- naive SpMM + validators
// reference spmm<L>, spmm.h:223-265: thread per row, first edge a plain product, the rest FMAs,
// empty rows left untouched.  Columns are walked in register tiles of 8 (any F, not a template).
__global__ __launch_bounds__(128) void k_spmm_naive(const int *__restrict__ ptr, const int *__restrict__ idx,
                                                   const float *__restrict__ val, const float *__restrict__ x,
                                                   float *__restrict__ y, int V, int F)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= V) return;
    const int beg = ptr[r], end = ptr[r + 1];
    if (beg == end) return;
    for (int c0 = 0; c0 < F; c0 += 8) {
        float ans[8];
        const int n = F - c0 < 8 ? F - c0 : 8;
        {
            const float v = val[beg];
            const float *xr = x + (size_t)idx[beg] * F + c0;
            for (int j = 0; j < n; ++j) ans[j] = v * xr[j];
        }
        for (int e = beg + 1; e < end; ++e) {
            const float v = val[e];
            const float *xr = x + (size_t)idx[e] * F + c0;
            for (int j = 0; j < n; ++j) ans[j] = __builtin_fmaf(v, xr[j], ans[j]);
        }
        for (int j = 0; j < n; ++j) y[(size_t)r * F + c0 + j] = ans[j];
    }
}

int launch_spmm_naive(const int *ptr, const int *idx, const float *val, const float *x, float *y, int V, int feat,
                      void *stream_v)
{
    hipStream_t stream = (hipStream_t)stream_v;
    if (V <= 0) return GNNAGG_OK;
    hipLaunchKernelGGL(k_spmm_naive, dim3(ceil_div(V, 128)), dim3(128), 0, stream, ptr, idx, val, x, y, V, feat);
    HIP_TRY(hipGetLastError());
    return GNNAGG_OK;
}

// reference validate2, spmm.h:11-21
__global__ void k_validate(const float *__restrict__ ref, const float *__restrict__ ans, int num, int *diff)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < num && fabsf((ref[t] - ans[t]) / ref[t]) > 1e-2f) atomicAdd(diff, 1);
}

// reference validateReordered, spmm.h:23-33
__global__ void k_validate_reordered(const float *__restrict__ ref, const float *__restrict__ ans,
                                     const int *__restrict__ map, int V, int F, int *diff)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < (long)V * F && fabsf(ref[t] - ans[(size_t)map[t / F] * F + t % F]) > 1e-2f) atomicAdd(diff, 1);
}

int launch_validate(const float *ref, const float *ans, int num, int *d_diff, void *stream_v)
{
    hipStream_t stream = (hipStream_t)stream_v;
    HIP_TRY(hipMemsetAsync(d_diff, 0, sizeof(int), stream));
    if (num > 0) hipLaunchKernelGGL(k_validate, dim3(ceil_div(num, 256)), dim3(256), 0, stream, ref, ans, num, d_diff);
    HIP_TRY(hipGetLastError());
    return GNNAGG_OK;
}

int launch_validate_reordered(const float *ref, const float *ans, const int *map, int V, int feat, int *d_diff,
                              void *stream_v)
{
    hipStream_t stream = (hipStream_t)stream_v;
    HIP_TRY(hipMemsetAsync(d_diff, 0, sizeof(int), stream));
    if ((long)V * feat > 0)
        hipLaunchKernelGGL(k_validate_reordered, dim3(ceil_div((long)V * feat, 256)), dim3(256), 0, stream, ref, ans,
                           map, V, feat, d_diff);
    HIP_TRY(hipGetLastError());
    return GNNAGG_OK;
}

// ------------------------------------------------------------------- dense combine GEMM (MFMA)
// C[M,N] = A[M,K] . B[K,N], all row-major fp32 -- the reference's matmul_NN (include/dense.h:4-23: cuBLAS
// Sgemm(T,T) + Sgeam transpose) and the dense half of aggr_gcn_nn (aggr_gcn.h:304-359).  Tall-skinny in this
// path (M = |V|, K = feat_in, N = feat_out <= a few hundred): HBM-bound on reading A once.
// One wavefront owns a 32x32 output tile and accumulates it with v_mfma_f32_32x32x2_f32 (f32 in / f32
// accumulate; bit-for-bit an ascending-k fmaf chain, so the result equals the oracle's chain exactly).
// A workgroup = 4 wavefronts = 128 rows x 32 columns; K is walked in chunks of 32 staged through LDS:
// A chunk with coalesced 128-byte row segments into a pitch-33 image (conflict-free operand reads:
// lane l reads row l&31, k = l>>5), B chunk as is (lane reads consecutive columns).
typedef float f32x16 __attribute__((ext_vector_type(16)));
static constexpr int kGemmRows = 128, kGemmCols = 32, kGemmKC = 32, kGemmPitch = 33;

__global__ __launch_bounds__(256) void k_dense_nn(const float *__restrict__ A, const float *__restrict__ B,
                                                  float *__restrict__ C, int M, int N, int K)
{
    __shared__ float As[kGemmRows * kGemmPitch];
    __shared__ float Bs[kGemmKC * kGemmCols];
    const int row0 = blockIdx.x * kGemmRows, col0 = blockIdx.y * kGemmCols;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    for (int k0 = 0; k0 < K; k0 += kGemmKC) {
        // stage A[row0 .. +128, k0 .. +32): thread t loads rows t/8 + 32*j, floats (t%8)*4 .. +4
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = (threadIdx.x >> 3) + 32 * j, kq = (threadIdx.x & 7) * 4;
            const int gr = row0 + r, gk = k0 + kq;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (gr < M) {
                const float *src = A + (size_t)gr * K + gk;
                if (gk + 3 < K && ((uintptr_t)src & 15) == 0) {
                    const float4 t = *reinterpret_cast<const float4 *>(src);
                    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) if (gk + q < K) v[q] = src[q];
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) As[r * kGemmPitch + kq + q] = v[q];
        }
        // stage B[k0 .. +32, col0 .. +32): 1024 floats, 4 per thread
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = threadIdx.x + 256 * j, kk = e >> 5, cc = e & 31;
            Bs[e] = (k0 + kk < K && col0 + cc < N) ? B[(size_t)(k0 + kk) * N + col0 + cc] : 0.0f;
        }
        __syncthreads();
        // all 16 operand pairs of the chunk into registers first, then 16 back-to-back MFMAs
        float av[kGemmKC / 2], bv[kGemmKC / 2];
#pragma unroll
        for (int t = 0; t < kGemmKC / 2; ++t) {
            av[t] = As[(wave * 32 + (lane & 31)) * kGemmPitch + 2 * t + (lane >> 5)];
            bv[t] = Bs[(2 * t + (lane >> 5)) * kGemmCols + (lane & 31)];
        }
#pragma unroll
        for (int t = 0; t < kGemmKC / 2; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[t], acc, 0, 0, 0);
        __syncthreads();
    }
    // C/D layout of 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const int col = col0 + (lane & 31);
    if (col < N) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = row0 + wave * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
            if (row < M) C[(size_t)row * N + col] = acc[reg];
        }
    }
}

// Tall-skinny variant for the aggregation widths (K <= 128, K % 4 == 0): every wavefront keeps its B operands -- the
// whole W[K, 32] column block, 64 VGPRs -- in registers for the life of the kernel and walks 32-row tiles of A on its own:
// 16 coalesced 16-byte loads per lane fetch the NEXT tile while the current one is multiplied; the tile passes through a
// per-wavefront LDS image (pitch K + 4: aligned ds_write_b128, operand reads two per bank) only to turn rows-over-lanes
// into the MFMA operand layout; no workgroup barrier anywhere.  Four 16x16 sub-tiles per tile, each the full ascending-k
// chain on v_mfma_f32_16x16x4_f32 (bit-exact as k_dense_nn).
static constexpr int kTallWaves = 2;  // wavefronts per workgroup (one 16.9 KB LDS image each at K = 128)

__global__ __launch_bounds__(64 * kTallWaves) void k_dense_nn_tall(const float *__restrict__ A, const float *__restrict__ B,
                                                                   float *__restrict__ C, int M, int N, int K, int ntiles)
{
    extern __shared__ float lds[];
    const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    const int pitch = K + 4, q4 = K >> 2;  // K % 4 == 0
    float *tile = lds + wave * 32 * pitch;
    const int col0 = blockIdx.y * 32;
    const int kq = lane >> 4, cl = lane & 15;
    // B operands: lane (c = lane % 16, k = lane / 16) of MFMA t holds W[4t + k][col]; two column halves
    float b0[32], b1[32];
#pragma unroll
    for (int t = 0; t < 32; ++t) {
        const int k = 4 * t + kq;
        b0[t] = (k < K && col0 + cl < N) ? B[(size_t)k * N + col0 + cl] : 0.0f;
        b1[t] = (k < K && col0 + 16 + cl < N) ? B[(size_t)k * N + col0 + 16 + cl] : 0.0f;
    }
    const int wstride = gridDim.x * kTallWaves;
    int t_idx = blockIdx.x * kTallWaves + wave;
    float4 pre[16];
    // float4 number e = lane + 64 i of a tile is (row e / q4, quad e % q4); stepping e by 64 advances (row, quad) by
    // (64 / q4, 64 % q4) with one carry -- no division in the loops
    const int step_r = 64 / q4, step_c = 64 - step_r * q4;
    const int r0 = lane / q4, c0 = lane - r0 * q4;
    auto fetch = [&](int ti) {
        int r = r0, c4 = c0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = ti * 32 + r;
            pre[i] = (r < 32 && row < M) ? *reinterpret_cast<const float4 *>(A + (size_t)row * K + 4 * c4)
                                         : make_float4(0.f, 0.f, 0.f, 0.f);
            r += step_r; c4 += step_c;
            if (c4 >= q4) { c4 -= q4; ++r; }
        }
    };
    if (t_idx < ntiles) fetch(t_idx);
    for (; t_idx < ntiles; t_idx += wstride) {
        {
            int r = r0, c4 = c0;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (r < 32) *reinterpret_cast<float4 *>(&tile[r * pitch + 4 * c4]) = pre[i];
                r += step_r; c4 += step_c;
                if (c4 >= q4) { c4 -= q4; ++r; }
            }
        }
        if (t_idx + wstride < ntiles) fetch(t_idx + wstride);  // travels during the MFMA chains below
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int rh = 0; rh < 2; ++rh) {
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            const float *arow = tile + (rh * 16 + cl) * pitch + kq;
            float av[32];
#pragma unroll
            for (int t = 0; t < 32; ++t) av[t] = 4 * t < K ? arow[4 * t] : 0.0f;
#pragma unroll
            for (int t = 0; t < 32; ++t) {
                if (4 * t < K) {  // wave-uniform
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], b0[t], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], b1[t], acc1, 0, 0, 0);
                }
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) {  // D layout: col = lane % 16, row = 4 * (lane / 16) + reg
                const int row = t_idx * 32 + rh * 16 + 4 * kq + v;
                if (row < M) {
                    if (col0 + cl < N) C[(size_t)row * N + col0 + cl] = acc0[v];
                    if (col0 + 16 + cl < N) C[(size_t)row * N + col0 + 16 + cl] = acc1[v];
                }
            }
        }
        __builtin_amdgcn_wave_barrier();  // the image is rewritten next iteration
    }
}

int launch_dense_nn(const float *A, const float *B, float *C, int M, int N, int K, void *stream_v)
{
    hipStream_t stream = (hipStream_t)stream_v;
    if (M <= 0 || N <= 0) return GNNAGG_OK;
    {
        static const int tall = getenv("GNNAGG_GEMM_TALL") ? atoi(getenv("GNNAGG_GEMM_TALL")) : 1;
        // measured against k_dense_nn (N = 32): K = 128: M = 300 k 48.2 vs 46.4 us, 600 k 92.4 vs 99.9, 1.2 M 169 vs 184,
        // 2.45 M 307 vs 352; K = 100, M = 2.45 M: 268 vs 363 (torch.mm 427); K = 64 loses at every M -> large M, wide K only
        if (tall && K > 64 && K <= 128 && (K & 3) == 0 && M >= 500000 && ((uintptr_t)A & 15) == 0) {
            const int ntiles = ceil_div(M, 32);
            const size_t lds = (size_t)kTallWaves * 32 * (K + 4) * sizeof(float);
            const int wgs = std::min(ceil_div(ntiles, kTallWaves), 256 * 4);
            hipLaunchKernelGGL(k_dense_nn_tall, dim3(wgs, ceil_div(N, 32)), dim3(64 * kTallWaves), lds, stream, A, B, C, M, N, K,
                               ntiles);
            HIP_TRY(hipGetLastError());
            return GNNAGG_OK;
        }
    }
    if (K <= 0) {
        HIP_TRY(hipMemsetAsync(C, 0, (size_t)M * N * sizeof(float), stream));
        return GNNAGG_OK;
    }
    const dim3 grid(ceil_div(M, kGemmRows), ceil_div(N, kGemmCols));
    hipLaunchKernelGGL(k_dense_nn, grid, dim3(256), 0, stream, A, B, C, M, N, K);
    HIP_TRY(hipGetLastError());
    return GNNAGG_OK;
}

// --------------------------------------------------------------------------------- CSR check
// The reference trusts its inputs (an out-of-range neighbor id is a silent out-of-bounds gather).  counts[0] = rows with
// ptr[r] > ptr[r+1], counts[1] = neighbor ids outside [0, num_cols).
__global__ void k_check_csr(const int *__restrict__ ptr, const int *__restrict__ idx, int V, int E, int num_cols, int *counts)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < V && ptr[t] > ptr[t + 1]) atomicAdd(&counts[0], 1);
    if (t < E && (idx[t] < 0 || idx[t] >= num_cols)) atomicAdd(&counts[1], 1);
}

int launch_check_csr(const int *ptr, const int *idx, int V, int E, int num_cols, int *d_counts, void *stream_v)
{
    hipStream_t stream = (hipStream_t)stream_v;
    HIP_TRY(hipMemsetAsync(d_counts, 0, 2 * sizeof(int), stream));
    const long n = std::max<long>(V, E);
    if (n > 0) hipLaunchKernelGGL(k_check_csr, dim3(ceil_div(n, 256)), dim3(256), 0, stream, ptr, idx, V, E, num_cols, d_counts);
    HIP_TRY(hipGetLastError());
    return GNNAGG_OK;
}

// ----------------------------------------------------------------------------- halo packing
// out[i,:] = x[ids[i],:]  -- send buffer of the halo all-to-all (gnnagg.h Section D)
template <int VEC, int GROUP>
__global__ __launch_bounds__(kBlock) void k_pack_rows(const float *__restrict__ x, const int *__restrict__ ids, int n,
                                                     int F, int ntiles, float *__restrict__ out)
{
    const int tile = blockIdx.x % ntiles;
    const int i = (blockIdx.x / ntiles) * (kBlock / GROUP) + threadIdx.x / GROUP;
    const int col = (tile * GROUP + (threadIdx.x & (GROUP - 1))) * VEC;
    if (i >= n || col >= F) return;
    const Pack<VEC> p = load_pack<VEC>(x + (size_t)ids[i] * F + col);
    store_pack<VEC>(out + (size_t)i * F + col, p.v);
}

int launch_pack_rows(const float *x, const int *ids, int n, int feat, float *out, void *stream_v)
{
    hipStream_t stream = (hipStream_t)stream_v;
    if (n <= 0) return GNNAGG_OK;
    const Geometry g = pick_geometry(feat, x, out, nullptr, feat);
    const int nb = ceil_div(n, kBlock / g.group) * g.ntiles;
#define CALL_PACK hipLaunchKernelGGL((k_pack_rows<VEC, GROUP>), dim3(nb), dim3(kBlock), 0, stream, x, ids, n, feat, g.ntiles, out);
    DISPATCH_GEOM(g, CALL_PACK)
#undef CALL_PACK
    HIP_TRY(hipGetLastError());
    return GNNAGG_OK;
}

}  // namespace gnnagg
