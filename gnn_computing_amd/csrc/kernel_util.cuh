#pragma once
// kernel_util.cuh -- shared device helpers of the hand-written CDNA4 (gfx950) kernels of the neighbor-aggregation hot
// path (agg_gcn.hip, agg_gat.hip, aux_kernels.hip).
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
#include <type_traits>

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
// neighbor gathers in flight per lane group: the default.  Round 1 (arxiv-shaped): 4 and 8 tie un-reordered (87.9 / 86.8 us), 16
// loses (96 us, register pressure), forcing 8 waves/SIMD at 8 gathers spills (137 us).  Round 2: WITH the locality reorder 4
// gathers (59 VGPRs, 8 waves per SIMD) beat 8 (76 VGPRs, 6 waves): 74.0 vs 77.9 us -- k_gcn_plan / k_gat_plan therefore
// take the batch size as a template parameter (4 on the 32- and 64-lane float4 geometries of the balanced / scheduled
// orders, this default elsewhere: long chains and narrow geometries want 8).
#ifndef KUNROLL
#define KUNROLL 8
#endif
static constexpr int kUnroll = KUNROLL;
static constexpr int kSegChunks = 16;  // chunks of a long row that one segment workgroup folds in LDS (plan kernels)

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

// Last step of a GCN / SAGE row: mean division, the join with what the row already holds (GNNAGG_FLAG_ACCUMULATE), ReLU.
// row_aux (gnnagg_set_row_aux; the row-partitioned step's two passes) changes two things: the mean divides by row_aux[row]
// -- the row's degree in the WHOLE graph -- instead of the edges this handle holds, so that local and halo parts can be
// added; and a max joins as max(old, new) only where row_aux[row] edges > 0 were already folded into y (else y = new).
template <int VEC, bool IS_MAX>
__device__ __forceinline__ void finish_gcn_row(float (&acc)[VEC], int own_deg, int row, const float *yold, int mean, int accumulate,
                                               int relu, const int *__restrict__ row_aux)
{
    if (mean) {
        const float dg = (float)(row_aux ? row_aux[row] : own_deg);
        if (dg > 0.0f) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] = acc[k] / dg;
        }
    }
    if (accumulate) {
        const Pack<VEC> old = load_pack<VEC>(yold);
        if (IS_MAX) {
            if (!row_aux || row_aux[row] > 0) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[k] = old.v[k] > acc[k] ? old.v[k] : acc[k];
            }
        } else {
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] = old.v[k] + acc[k];
        }
    }
    if (relu) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = acc[k] > 0.0f ? acc[k] : 0.0f;
    }
}

// Store of a lane's VEC results when the destination's alignment class (`avec` = 4 / 2 / 1: 16-, 8- or 4-byte aligned
// rows) or the number of valid columns (`nvalid`, the ragged last tile) is below VEC: the 2-D blocked mode computes on
// 16-byte lanes (re-tiled X) whatever the caller's row pitch is.
template <int VEC>
__device__ __forceinline__ void store_pack_any(float *p, const float (&a)[VEC], int nvalid, int avec)
{
    if (nvalid >= VEC && avec >= VEC) {
        store_pack<VEC>(p, a);
    } else if (VEC == 4 && nvalid >= 4 && avec >= 2) {
        *reinterpret_cast<float2 *>(p) = make_float2(a[0], a[1]);
        *reinterpret_cast<float2 *>(p + 2) = make_float2(a[2], a[3]);
    } else {
#pragma unroll
        for (int k = 0; k < VEC; ++k)
            if (k < nvalid) p[k] = a[k];
    }
}

// Output rows are written once and never re-read by this kernel: a write-through (sc1) store leaves the XCD's L2
// to the gathered feature rows instead of parking 87 MB of results in it.  Buffer store so the cache bits can be
// given (aux 16 = sc1); `yoff` is the element offset from `ybase` (callers guarantee the byte offset fits 31 bits).
#ifndef GNNAGG_WT_AUX
#define GNNAGG_WT_AUX 16
#endif
// AUX: cache-policy bits of the buffer store (gfx950: 1 = sc0, 2 = nt, 16 = sc1).  sc1 (write-through to device scope) for
// results and for the hub scratch other XCDs read in the same launch; nt (streaming) for the partial rows of the 2-D
// blocked order, which only a later launch reads: they then do not displace the X slice the L2 is there to hold
// (reddit-shaped F = 602: 15.40 -> 14.97 ms, GAT 8 x 32: 7.84 -> 7.56 ms; nt + sc1: 15.15 / 7.69).
template <int VEC, int AUX = GNNAGG_WT_AUX>
__device__ __forceinline__ void store_pack_wt(float *ybase, unsigned nbytes, size_t yoff, const float (&a)[VEC])
{
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(ybase, 0, (int)nbytes, 0x00020000);
    const int voff = (int)(yoff * sizeof(float));
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    if constexpr (VEC == 4) {
        u4 v = {__float_as_uint(a[0]), __float_as_uint(a[1]), __float_as_uint(a[2]), __float_as_uint(a[3])};
        __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, voff, 0, AUX);
    } else if constexpr (VEC == 2) {
        u2 v = {__float_as_uint(a[0]), __float_as_uint(a[1])};
        __builtin_amdgcn_raw_buffer_store_b64(v, rsrc, voff, 0, AUX);
    } else {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(a[0]), rsrc, voff, 0, AUX);
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

// 2-D blocked mode (source range x column tile): the linear work index runs TILE-major, L = tile * item_blocks + ib, so a
// contiguous XCD range walks ONE column tile of ONE source range after the other -- the slice of X its L2 has to hold is
// (rows of a range) x (tile bytes).  xr.first / xr.count are cut over L.  Returns ib * ntiles + tile like logical_block.
__device__ __forceinline__ int logical_block_tile_major(int b, int item_blocks, int ntiles, const XcdRanges &xr)
{
    const int xcd = b & 7, k = b >> 3;
    if (k >= xr.count[xcd]) return -1;
    const int L = xr.first[xcd] + k;
    const int tile = L / item_blocks;
    return (L - tile * item_blocks) * ntiles + tile;
}

// A/B switches for the DPP forms of the id / value broadcast (measured: DESIGN.md section 4).
#ifndef GNNAGG_DPP_CHAIN
#define GNNAGG_DPP_CHAIN 0
#endif
#ifndef GNNAGG_DPP_SPAN_GCN
#define GNNAGG_DPP_SPAN_GCN 0
#endif

// Compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{})
template <int N, class Fn>
__device__ __forceinline__ void static_for(Fn &&f)
{
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// Value of lane SRC (a compile-time index) of every lane group, in all lanes of the group.  16-lane groups are exactly the
// DPP rows: row_newbcast is a full-rate VALU move with no LDS round trip -- ds_bpermute, what __shfl compiles to, goes
// through the LDS pipe and puts ~100 cycles into the id -> address -> gather chain.  Used by the GAT span kernel (10.2 ->
// 8.7 ms on the reddit-shaped 8 x 32 case); on the GCN kernels the unrolled form costs registers and measured slower
// (arxiv-shaped headline 77 -> 103 us, reddit-shaped F = 602 15.9 -> 16.1 ms): GNNAGG_DPP_CHAIN / GNNAGG_DPP_SPAN_GCN stay 0.
template <int GROUP, int SRC>
__device__ __forceinline__ int group_bcast(int v)
{
    if constexpr (GROUP == 16) return __builtin_amdgcn_mov_dpp(v, 0x150 + SRC, 0xf, 0xf, true);  // every lane is written
    else return __shfl(v, SRC, GROUP);
}
template <int GROUP, int SRC>
__device__ __forceinline__ float group_bcast(float v)
{
    return __int_as_float(group_bcast<GROUP, SRC>(__float_as_int(v)));
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
template <int VEC, int GROUP, bool IS_MAX, int UNROLL = kUnroll>
__device__ __forceinline__ void chain_edges(float (&acc)[VEC], int beg, int end, int lane, bool col_ok,
                                            const int *__restrict__ idx, const float *__restrict__ val,
                                            const float *__restrict__ xcol, int F)
{
    if constexpr (GROUP >= 16 && GNNAGG_DPP_CHAIN) {
        // Windows of 16 edges, one copy per 16-lane DPP row of the group (lanes l and l + 16 load the same id: one request):
        // the (id, value) of edge cb + u then reaches every lane of the group by row_newbcast -- a VALU move instead of the
        // ds_bpermute round trip through the LDS pipe (arxiv-shaped headline: see DESIGN.md section 4).  Same edge order,
        // same bits.
        const int l16 = lane & 15;
        int my_s = 0;
        float my_w = 1.0f;
        if (beg + l16 < end) {
            my_s = idx[beg + l16];
            if (val) my_w = val[beg + l16];
        }
        for (int cb = beg; cb < end; cb += 16) {
            int nx_s = 0;
            float nx_w = 1.0f;
            if (cb + 16 + l16 < end) {
                nx_s = idx[cb + 16 + l16];
                if (val) nx_w = val[cb + 16 + l16];
            }
            const int n = end - cb < 16 ? end - cb : 16;
            static_for<16 / UNROLL>([&](auto bc) {
                constexpr int J = decltype(bc)::value * UNROLL;
                if (J < n) {
                    int s[UNROLL];
                    float w[UNROLL];
                    Pack<VEC> xv[UNROLL];
                    static_for<UNROLL>([&](auto uc) {
                        constexpr int u = decltype(uc)::value;
                        s[u] = group_bcast<16, J + u>(my_s);
                        w[u] = group_bcast<16, J + u>(my_w);
                    });
#pragma unroll
                    for (int u = 0; u < UNROLL; ++u)
                        if (J + u < n && col_ok) xv[u] = load_pack<VEC>(xcol + (size_t)s[u] * F);
#pragma unroll
                    for (int u = 0; u < UNROLL; ++u)
                        if (J + u < n && col_ok) {
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
            });
            my_s = nx_s;
            my_w = nx_w;
        }
        return;
    }
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
        for (int j = 0; j < n; j += UNROLL) {
            int s[UNROLL];
            float w[UNROLL];
            Pack<VEC> xv[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                s[u] = __shfl(my_s, j + u, GROUP);
                w[u] = __shfl(my_w, j + u, GROUP);
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
                if (j + u < n && col_ok) xv[u] = load_pack<VEC>(xcol + (size_t)s[u] * F);
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
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

// exp(leaky_relu(a_dst + a_src)): the un-normalised attention weight of an edge (aggr_gat.h:138-143)
__device__ __forceinline__ float edge_weight(float a_dst, float a_src, float slope)
{
    // reference aggr_gat.h:138-143: exp(max(s, s*slope)), no max-subtraction
    const float sc = a_dst + a_src;
    const float l = sc * slope;
    return expf(sc > l ? sc : l);
}

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

// sum over the lanes of a group (xor butterfly: a fixed association, every lane gets the total)
template <int GROUP>
__device__ __forceinline__ float group_sum(float v)
{
#pragma unroll
    for (int m = GROUP / 2; m > 0; m >>= 1) v += __shfl_xor(v, m, GROUP);
    return v;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

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

// Launch-site state that belongs to a DEVICE, not to the process (a process may drive several): "this function attribute has been
// set here" flags and the CU count.  (A relaxed race sets an attribute twice, which is harmless.)
struct OncePerDevice {
    unsigned long long mask = 0;   // bit d: done on device d (devices >= 64: never cached)
    int dev = -1;
    bool first()
    {
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { dev = -1; return true; }
        return ((__atomic_load_n(&mask, __ATOMIC_RELAXED) >> dev) & 1ull) == 0;
    }
    void done() { if (dev >= 0) __atomic_fetch_or(&mask, 1ull << dev, __ATOMIC_RELAXED); }
};
static int device_cu_count()
{
    static int cus[64] = {};
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cus[dev] > 0) return cus[dev];
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    return cus[dev] = n;
}

// One word of device memory per DEVICE for the probe instantiations' never-taken store (a process-wide pointer allocated on
// the first device would be dereferenced by kernels of handles that live on another one).
static unsigned *device_probe_sink()
{
    static unsigned *sinks[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (!sinks[dev] && hipMalloc((void **)&sinks[dev], sizeof(unsigned)) != hipSuccess) sinks[dev] = nullptr;
    return sinks[dev];
}

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

// Tile-major variant: the work list is (tile 0: all item blocks)(tile 1: all item blocks)...; 8 contiguous ranges of about
// equal cost over that list.  first/count are in linear (tile * item_blocks + ib) units; returns the longest range.
static int fill_xcd_ranges_tile_major(const long *cost_prefix, int n_items, int items_per_block, int item_blocks, int ntiles,
                                      XcdRanges &xr)
{
    const long per_tile = cost_prefix[n_items];
    const long total = per_tile * ntiles;
    long start = 0;
    int longest = 0;
    for (int x = 0; x < 8; ++x) {
        long stop;
        if (x == 7 || per_tile == 0) {
            stop = x == 7 ? (long)item_blocks * ntiles : start;
        } else {
            const long want = total * (x + 1) / 8;
            const long tile = std::min<long>(want / per_tile, ntiles - 1);
            const long rem = want - tile * per_tile;
            int lo = 0, hi = item_blocks;
            while (lo < hi) {
                const int mid = (lo + hi) / 2;
                const long c = cost_prefix[std::min((long)mid * items_per_block, (long)n_items)];
                if (c < rem) lo = mid + 1; else hi = mid;
            }
            stop = std::max(start, tile * item_blocks + lo);
        }
        xr.first[x] = (int)start;
        xr.count[x] = (int)(stop - start);
        longest = std::max(longest, (int)(stop - start));
        start = stop;
    }
    return longest;
}

}  // namespace gnnagg
