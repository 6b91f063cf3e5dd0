// agg_gcn.hip -- GCN / GraphSAGE aggregation kernels (items, balanced plan, fused dense epilogue, rows-mode long rows)
// and their launchers.  Shared helpers: kernel_util.cuh; the ordered combine: combine.cuh.
#include "combine.cuh"

namespace gnnagg {

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
//    segment is written straight to Y; only rows with several segments (hubs) go through scratch: the last of their
//    segment workgroups to arrive folds them (hub_arrive_and_fold), or k_combine when that is switched off.
//  * the remaining blocks: GPB short rows each (deg <= chunk, empty rows included), one lane group per
//    row, descriptor {beg,end,row} fetched with ONE 16-byte load; XCD-aware work-balanced block ranges.

struct PlanArgs {
    const int4 *t0;  // {beg, end, row, -}
    const int4 *t1;  // {beg, end, dest (>=0 row, <0 ~scratch slot), -}
    const int *idx;
    const float *val;
    const float *x;
    float *y;
    float *partial;
    int n0, n1, feat, ntiles, chunk, mean, remap, nblocks0;
    int accumulate;  // 1: y += result (rows without edges are left untouched); mean / max with row_aux only
    const int *row_aux;  // gnnagg_set_row_aux (finish_gcn_row); nullptr otherwise
    int relu;        // 1: y = max(result, 0)
    int wt;          // 1: write-through (sc1) stores of the short-row results
    unsigned ybytes;
    // hubs (rows with several segments) folded by the last segment workgroup to arrive; hub_count == nullptr: k_combine
    const int *slot_hub, *mrow_ptr, *mrow_id, *row_ptr;
    int *hub_count;
    int hub_count_stride;
    unsigned partial_bytes;
    // addressing of X and of the partial scratch (floats): row-major defaults (xpitch = ppitch = feat, tile strides = the lane
    // group's column span) or the 2-D blocked mode's tile-major images (TileSpec); tile_major: block order of that mode
    int xpitch, ppitch, yvec, tile_major, item_blocks;
    long x_tile_stride, p_tile_stride;
    unsigned ptile_bytes;  // bytes of one tile of the partial scratch when that fits a buffer descriptor, else 0
    unsigned *probe_sink;  // PROBE instantiation only
    XcdRanges xr;
};

// Gather probe (bench.py's measured ceiling): everything chain_edges loads -- ids, values, feature segments, same
// addresses, same batching -- consumed with integer XORs instead of the dependent FMA chain.
template <int VEC, int GROUP, int UNROLL = kUnroll>
__device__ __forceinline__ unsigned probe_edges(int beg, int end, int lane, bool col_ok, const int *__restrict__ idx,
                                                const float *__restrict__ val, const float *__restrict__ xcol, int F)
{
    unsigned sig = 0;
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
            Pack<VEC> xv[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                s[u] = __shfl(my_s, j + u, GROUP);
                sig ^= __float_as_uint(__shfl(my_w, j + u, GROUP));
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
                if (j + u < n && col_ok) xv[u] = load_pack<VEC>(xcol + (size_t)s[u] * F);
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
                if (j + u < n && col_ok) {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) sig ^= __float_as_uint(xv[u].v[k]);
                }
        }
        my_s = nx_s;
        my_w = nx_w;
    }
    return sig;
}

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
    // Publication protocol = the write-through form of the inter-workgroup hand-off (cdna_hip_programming.md Guideline 16,
    // recipe R1; MI355X_MICROARCH.md "Valid forms"): payload stored with sc1 (write-through, leaves this XCD's L2), EVERY
    // storing wave drains its stores (asm wait: the one form the compiler can neither drop nor move memory operations
    // across), workgroup barrier, ONE lane's agent-scope atomic on the arrival counter; the consumer -- the workgroup whose
    // atomic returns nseg - 1 -- reads the payload with sc1 loads (L1-bypassing; valid without an acquire fence because the
    // producers stored sc1).  Placement-independent: nothing assumes which XCD a segment workgroup runs on.
    if (grp == 0 && col_ok) store_pack_wt<VEC>(a.partial, a.partial_bytes, (size_t)slot * F + col, acc);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __shared__ int s_hub;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int m = a.slot_hub[slot];
        const int nseg = a.mrow_ptr[m + 1] - a.mrow_ptr[m];
        int *cnt = a.hub_count + (size_t)m * a.hub_count_stride + tile;
        const int old = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // everybody is in: zero for the next launch (ordered behind this launch's adds by the kernel boundary)
        if (old == nseg - 1) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
        finish_gcn_row<VEC, IS_MAX>(acc, a.mean ? a.row_ptr[row + 1] - a.row_ptr[row] : 1, row, a.y + (size_t)row * F + col, a.mean,
                                    a.accumulate, a.relu, a.row_aux);
        store_pack<VEC>(a.y + (size_t)row * F + col, acc);
    }
    return true;
}

#ifdef GNNAGG_PLAN_WAVES  // A/B: ask the register allocator for this many waves per SIMD (default: what 76 VGPRs give, 6)
#define PLAN_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(GNNAGG_PLAN_WAVES, GNNAGG_PLAN_WAVES)))
#else
#define PLAN_WAVES_ATTR
#endif
// UNROLL: row gathers issued per batch before the first FMA.  8 by default; 4 for the 32- and 64-lane float4 geometries (F > 64)
// of the balanced / scheduled orders: 59 instead of 76 VGPRs, 8 instead of 6 waves per SIMD -- the arxiv-shaped headline
// with the locality reorder 77.9 -> 74.0 us (without reorder, and at F = 100: unchanged); the canonical rows mode keeps 8
// (its chains are long: 182 -> 209 us with 4), and so do the narrow geometries (F = 32 balanced: 28.6 -> 36.3 us with 4).
template <int VEC, int GROUP, bool IS_MAX, bool PROBE = false, int UNROLL = kUnroll>
__global__ __launch_bounds__(block_of<GROUP>()) PLAN_WAVES_ATTR void k_gcn_plan(const PlanArgs a)
{
    constexpr int GPB = block_of<GROUP>() / GROUP;
    const int F = a.feat;
    const int lane = threadIdx.x & (GROUP - 1);
    const int grp = (int)threadIdx.x / GROUP;
    const int nb1 = a.n1 * a.ntiles;
    if (PROBE && (int)blockIdx.x < nb1) {  // segments: the chunks' loads, nothing else
        const int tile = (int)blockIdx.x % a.ntiles;
        const int4 d = a.t1[(int)blockIdx.x / a.ntiles];
        const int col = (tile * GROUP + lane) * VEC;
        const int nch = (d.y - d.x + a.chunk - 1) / a.chunk;
        unsigned sig = 0;
        for (int c = grp; c < nch; c += GPB) {
            const int cb = d.x + c * a.chunk;
            const int ce = cb + a.chunk < d.y ? cb + a.chunk : d.y;
            sig ^= probe_edges<VEC, GROUP, UNROLL>(cb, ce, lane, col < F, a.idx, a.val, a.x + col, F);
        }
        if (sig == 0x9e3779b9u) a.probe_sink[0] = sig;  // practically never: keeps the loads alive
        return;
    }
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
            chain_edges<VEC, GROUP, IS_MAX, UNROLL>(acc, cb, ce, lane, col_ok, a.idx, a.val, a.x + col, F);
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
            finish_gcn_row<VEC, IS_MAX>(acc, d.y - d.x, d.z, a.y + (size_t)d.z * F + col, a.mean, a.accumulate, a.relu, a.row_aux);
            store_pack<VEC>(a.y + (size_t)d.z * F + col, acc);
        } else {
            store_pack<VEC>(a.partial + (size_t)(~d.z) * F + col, acc);
        }
        return;
    }
    const int b = a.tile_major ? logical_block_tile_major((int)blockIdx.x - nb1, a.item_blocks, a.ntiles, a.xr)
                               : logical_block((int)blockIdx.x - nb1, a.nblocks0, a.ntiles, a.remap, a.xr);
    if (b < 0) return;
    const int tile = b % a.ntiles;
    const int item = (b / a.ntiles) * GPB + grp;
    if (item >= a.n0) return;
    const int col = (tile * GROUP + lane) * VEC;
    const bool col_ok = col < F;
    const int4 d = a.t0[item];
    const float *__restrict__ xcol = a.x + (size_t)tile * a.x_tile_stride + lane * VEC;
    if constexpr (PROBE) {
        const unsigned sig = probe_edges<VEC, GROUP, UNROLL>(d.x, d.y, lane, col_ok, a.idx, a.val, xcol, a.xpitch);
        if (sig == 0x9e3779b9u) a.probe_sink[0] = sig;
        return;
    }
    if (a.accumulate && !a.relu && d.x == d.y) return;  // y += 0
    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = IS_MAX ? -INFINITY : 0.0f;
    chain_edges<VEC, GROUP, IS_MAX, UNROLL>(acc, d.x, d.y, lane, col_ok, a.idx, a.val, xcol, a.xpitch);
    if (!col_ok) return;
    if (d.z < 0) {  // one of several groups of its row (source-partitioned order): raw partial to its scratch slot
        // write-through when a descriptor can cover the tile: the partials are read back by k_combine only, and must not
        // push the X slice out of this XCD's L2
        const size_t poff = (size_t)(~d.z) * a.ppitch + lane * VEC;
        if (a.ptile_bytes) store_pack_wt<VEC>(a.partial + (size_t)tile * a.p_tile_stride, a.ptile_bytes, poff, acc);
        else store_pack<VEC>(a.partial + (size_t)tile * a.p_tile_stride + poff, acc);
        return;
    }
    if (d.x == d.y) {  // no edges: 0 (an accumulating run that comes here applies the ReLU to what the row holds)
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = 0.0f;
        finish_gcn_row<VEC, false>(acc, 1, d.z, a.y + (size_t)d.z * F + col, 0, a.accumulate, a.relu, nullptr);
    } else {
        finish_gcn_row<VEC, IS_MAX>(acc, d.y - d.x, d.z, a.y + (size_t)d.z * F + col, a.mean, a.accumulate, a.relu, a.row_aux);
    }
    if (a.yvec < VEC || F - col < VEC) store_pack_any<VEC>(a.y + (size_t)d.z * F + col, acc, F - col, a.yvec);
    else if (a.wt) store_pack_wt<VEC>(a.y, a.ybytes, (size_t)d.z * F + col, acc);
    else store_pack<VEC>(a.y + (size_t)d.z * F + col, acc);
}

// ------------------------------------------------- aggregation with the dense combine as its epilogue
// transformed[V,N] = (A . X)[V,K] . W[K,N] in one pass (reference aggr_gcn_nn, aggr_gcn.h:304-359, called by
// run_with_nn :491-499).  The aggregation spreads the K columns of a row over the lanes of a group while the matrix
// cores want rows across lanes, so finished rows meet in LDS: a workgroup aggregates kNnRows = 16 short rows (16/GPB passes
// of the plan kernel's descriptor path), stages them as a [rows][K] tile (pitch K + 4: aligned 16-byte row stores, operand reads two per bank),
// and after ONE barrier its 4 wavefronts each take 16x16 output sub-tiles and run the full-K chain on
// v_mfma_f32_16x16x4_f32 -- f32 in / f32 accumulate, an ascending-k fmaf chain, so the result is bit-for-bit the
// separate GEMM's (and the oracle's).  W (K*N*4 bytes, 16 KB at 128x32) is read through L1/L2, not staged.
// Unlike the reference (partial . W added with atomics per neighbor group) W is applied to the FINAL row: rows that
// are folded from several chunks get their product where the fold ends (segment workgroup, hub fold, k_combine: one fmaf
// chain per output column, row_times_weight); the rows-mode long rows from k_dense_rows after the join.
#ifndef NN_ROWS
#define NN_ROWS 16
#endif
static constexpr int kNnRows = NN_ROWS;  // short rows a workgroup of the fused kernel aggregates and multiplies

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
static constexpr int kLongBlock = 512;   // hub rows: seven gather wavefronts, 114 KB of LDS -- one workgroup per CU
// Rows between the two (`medium`: too long for one lane group's latency-bound walk, far too many to give each a CU) take the same
// kernel at 128 threads: ONE gather wavefront (64 neighbor segments per round), 16.5 KB of LDS, nine workgroups per CU.
static constexpr int kMediumBlock = 128;

struct RowsLongArgs {
    const int4 *r1;  // {beg, end, row, -}
    const int *idx;
    const float *val;
    const float *x;
    float *y;
    int n1, feat, ntiles, mean, relu;   // ntiles: column tiles of TW floats
    // GAT flavour (reference aggr_gat, aggr_gat.h:116-164): the edge weight is exp(leaky(att[row,h,0] + att[src,h,1]))
    // computed by the gathering lanes; the consumer also runs the denominator chain.  Needs dhead % 32 == 0 so
    // that a 32-column tile lies inside one head.
    const float *att;
    int heads, dhead;
    float slope;
};

static constexpr int kLongU = 8;                            // neighbors per gather group per round

template <int VEC, int BLOCK, int TW = 32, int NC = 1>
constexpr int long_round_edges() { return ((BLOCK - 64 * NC) / (TW / VEC)) * kLongU; }   // (the first NC wavefronts only consume)

// TW = 64 (hub form, GCN flavours): the consumer's upper 32 lanes run chains too -- the same instruction stream serves 64 columns, so a
// launch with many hub rows (bound by the consumers' 12 cycles per step, one consumer per CU) needs half the workgroups.  A single
// row gets no faster (its chain is as long) and has half the gather wavefronts working for it: the launcher takes TW = 64 only where
// the (row, tile) items outnumber the CUs.
// NC = 2 (GCN flavours): TWO consumer wavefronts take the round's 64-step batches in turn.  A consumer reads its batch from LDS into
// registers while the other one runs the chain, then waits for the accumulators -- {value, batches done} pairs in LDS, one 8-byte slot per
// lane -- and runs its 64 steps register-fed: the ds_read_b128 instructions (14 issue cycles each, 7 of the 12 cycles of a step) leave the
// chain's critical path; what remains is the FMA (5.1) and the hand-over once per 64 steps (~300 cycles: LDS write, poll, LDS read).
// arxiv-shaped, the 15 k-edge row: 95 -> 80 us alone, 99 -> 85 us beside the short rows; rows mode 122.5 -> 114 us.  With 64-column
// tiles (the throughput-bound launches) the same form loses -- products-shaped 9.1 -> 10.4 ms -- and is not instantiated.
static constexpr int kLongBlock2 = 576;   // 2 consumers + 7 gather wavefronts: rounds of 448 edges = 7 batches
template <int VEC, bool IS_MAX, bool IS_GAT, int BLOCK = kLongBlock, int TW = 32, int NC = 1>
__global__ __launch_bounds__(BLOCK) void k_gcn_rows_long(const RowsLongArgs a)
{
    constexpr int GL = TW / VEC;                 // lanes of one gather group: GL * VEC = TW columns = 128 (256) bytes
    constexpr int NG = (BLOCK - 64 * NC) / GL;   // gather groups per workgroup
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
    float *stage0 = lds, *stage1 = lds + RE * TW, *wst0 = lds + 2 * RE * TW, *wst1 = wst0 + RE;
    const int F = a.feat;
    const int tile = (int)blockIdx.x % a.ntiles;
    const int4 d = a.r1[(int)blockIdx.x / a.ntiles];
    const int nrounds = (d.y - d.x + RE - 1) / RE;
    const int head = IS_GAT ? (tile * TW) / a.dhead : 0;
    if constexpr (NC == 2) {
      if (threadIdx.x < 128) {
        static_assert(NC == 1 || (!IS_GAT && RE % 64 == 0), "two consumers: GCN flavours, rounds of whole 64-step batches");
        const int cw = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);   // which consumer
        const int c = (int)threadIdx.x & 63, cc = c & (TW - 1);   // (lanes past the tile shadow a real column; only `consumer` lanes store)
        const bool consumer = c < TW && tile * TW + c < F;
        unsigned long long *slot = reinterpret_cast<unsigned long long *>(wst1 + RE) + c;
        if (cw == 0) __hip_atomic_store(slot, (unsigned long long)__float_as_uint(IS_MAX ? -INFINITY : 0.0f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        int gb = 0;   // batches of 64 steps so far, over all rounds: batch gb belongs to consumer gb & 1
        for (int r = 0; r < nrounds; ++r) {
            __syncthreads();  // round r is staged (round 0: and the slots are initialised)
            const float *stage = (r & 1) ? stage1 : stage0, *wst = (r & 1) ? wst1 : wst0;
            const int base = d.x + r * RE;
            const int n = d.y - base < RE ? d.y - base : RE;
            const int nb = (n + 63) >> 6;
            for (int i = 0; i < nb; ++i, ++gb) {
                if ((gb & 1) != cw) continue;
                const int k = i << 6;
                const int nk = n - k < 64 ? n - k : 64;
                float4 xs[16], ws[16];   // (reads past the round's last edge stay inside the stage buffers: RE is a multiple of 64)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int kq = (k >> 2) + q;
                    xs[q] = *reinterpret_cast<const float4 *>(&stage[(kq * TW + (cc ^ ((kq >> 1) & 3))) * 4]);
                    ws[q] = *reinterpret_cast<const float4 *>(&wst[k + 4 * q]);
                }
                unsigned long long v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                while ((int)(v >> 32) != gb) {
#ifdef GNNAGG_EXP_POLL_SLEEP
                    __builtin_amdgcn_s_sleep(GNNAGG_EXP_POLL_SLEEP);
#endif
                    v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                float acc = __uint_as_float((unsigned)v);
                auto step = [&](float x1, float w1) {
                    if (IS_MAX) {
                        const float p = x1 * w1;
                        acc = p > acc ? p : acc;
                    } else {
                        acc = __builtin_fmaf(x1, w1, acc);
                    }
                };
                if (nk == 64) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        step(xs[q].x, ws[q].x); step(xs[q].y, ws[q].y); step(xs[q].z, ws[q].z); step(xs[q].w, ws[q].w);
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        if (4 * q < nk) step(xs[q].x, ws[q].x);
                        if (4 * q + 1 < nk) step(xs[q].y, ws[q].y);
                        if (4 * q + 2 < nk) step(xs[q].z, ws[q].z);
                        if (4 * q + 3 < nk) step(xs[q].w, ws[q].w);
                    }
                }
                if (r == nrounds - 1 && i == nb - 1) {
                    if (consumer) {
                        if (a.mean) acc = acc / (float)(d.y - d.x);
                        if (a.relu) acc = acc > 0.0f ? acc : 0.0f;
                        a.y[(size_t)d.z * F + tile * TW + c] = acc;
                    }
                } else {
                    __hip_atomic_store(slot, ((unsigned long long)(unsigned)(gb + 1) << 32) | __float_as_uint(acc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
        return;
      }
    }
    if (NC == 1 && threadIdx.x < 64) {
        // ---- consumer wavefront: lane c < 32 owns column tile*32 + c and runs its chain from LDS in edge order
        const int c = (int)threadIdx.x;
        const bool consumer = c < TW && tile * TW + c < F;
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
                    xs[q] = *reinterpret_cast<const float4 *>(&stage[(((k >> 2) + q) * TW + (c ^ ((q >> 1) & 3))) * 4]);
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
            if (n - k0 >= 16) {   // rounds that are not a multiple of 32 edges (RE = 112 at TW = 64 with 2-float lanes): one 16-step batch
                float4 xs[4], ws[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int kq = (k0 >> 2) + q;
                    xs[q] = *reinterpret_cast<const float4 *>(&stage[(kq * TW + (c ^ ((kq >> 1) & 3))) * 4]);
                    ws[q] = *reinterpret_cast<const float4 *>(&wst[k0 + 4 * q]);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    step(xs[q].x, ws[q].x);
                    step(xs[q].y, ws[q].y);
                    step(xs[q].z, ws[q].z);
                    step(xs[q].w, ws[q].w);
                }
                k0 += 16;
            }
            for (; k0 < n; ++k0) step(stage[((k0 >> 2) * TW + (c ^ ((k0 >> 3) & 3))) * 4 + (k0 & 3)], wst[k0]);
        }
        if (consumer) {
            if (IS_GAT) acc = acc / den;  // aggr_gat.h:163 (rows here are never empty)
            else if (a.mean) acc = acc / (float)(d.y - d.x);
            if (!IS_GAT && a.relu) acc = acc > 0.0f ? acc : 0.0f;
            a.y[(size_t)d.z * F + tile * TW + c] = acc;
        }
        return;
    }
    // ---- gather wavefronts: group g fetches the 128-byte tile segments of edges base + g*U .. +U of every round
    const int t = (int)threadIdx.x - 64 * NC;
    const int g = t / GL, lane = t & (GL - 1);
    const int col = tile * TW + lane * VEC;
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
                    *reinterpret_cast<float4 *>(&stage[(kq * TW + ((lane * VEC + j) ^ swz)) * 4]) =
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

static int launch_combine_gcn(const GcnLaunch &L, const Geometry &g, bool is_max, hipStream_t stream,
                              const float *nn_weight = nullptr, float *nn_out = nullptr, int nn_cols = 0,
                              const TileSpec *tile = nullptr)
{
    if (L.wl.n_mrows > 0) {
        CombineArgs c;
        combine_strides(c, L.feat, g, tile);
        c.mrow_id = L.wl.mrow_id; c.mrow_ptr = L.wl.mrow_ptr; c.row_ptr = L.row_ptr; c.partial = L.partial;
        c.partial_den = nullptr; c.y = L.y; c.n_mrows = L.wl.n_mrows; c.feat = L.feat; c.ntiles = g.ntiles;
        c.heads = 1; c.dhead = L.feat; c.mean = L.reduce == GNNAGG_REDUCE_MEAN;
        c.accumulate = L.accumulate; c.relu = L.relu; c.row_aux = L.row_aux;
        c.nn_weight = nn_weight; c.nn_out = nn_out; c.nn_cols = nn_cols;
        c.big_rows = L.wl.big_rows; c.n_big = L.wl.n_big;
        c.nblocks_small = ceil_div(c.n_mrows, kBlock / g.group) * g.ntiles;
        const int nb_big = c.n_big * g.ntiles;
#define CALL_COMB                                                                                           \
        if (is_max) hipLaunchKernelGGL((k_combine<VEC, GROUP, true, false, false>), dim3(c.nblocks_small), dim3(kBlock), 0, stream, c);  \
        else        hipLaunchKernelGGL((k_combine<VEC, GROUP, false, false, false>), dim3(c.nblocks_small), dim3(kBlock), 0, stream, c); \
        if (nb_big > 0) {                                                                                   \
            if (is_max) hipLaunchKernelGGL((k_combine<VEC, GROUP, true, false, true>), dim3(nb_big), dim3(kBlock), 0, stream, c);  \
            else        hipLaunchKernelGGL((k_combine<VEC, GROUP, false, false, true>), dim3(nb_big), dim3(kBlock), 0, stream, c); \
        }
        DISPATCH_GEOM(g, CALL_COMB)
#undef CALL_COMB
        HIP_TRY(hipGetLastError());
    }
    return GNNAGG_OK;
}

int launch_dense_nn(const float *A, const float *B, float *C, int M, int N, int K, void *stream_v);

int launch_dense_rows(const int *rows, int n_rows, const float *Y, const float *W, float *out, int K, int N, void *stream_v)
{
    hipStream_t stream = (hipStream_t)stream_v;
    if (n_rows <= 0 || N <= 0) return GNNAGG_OK;
    if ((size_t)K * sizeof(float) > 60 * 1024) return fail(GNNAGG_ERR_ARG, "dense_rows: feature length too large");
    hipLaunchKernelGGL(k_dense_rows, dim3(n_rows), dim3(kBlock), (size_t)K * sizeof(float), stream, rows, Y, W, out, K, N);
    HIP_TRY(hipGetLastError());
    return GNNAGG_OK;
}

// geometry of a 2-D blocked launch: 16-byte lanes over tiles of tile_w floats
static Geometry tile_geometry(const TileSpec &t, int feat) { return {4, t.tile_w / 4, (feat + t.tile_w - 1) / t.tile_w}; }

int launch_gcn_plan(const GcnPlanLaunch &L, void *stream_v)
{
    hipStream_t stream = (hipStream_t)stream_v;
    if (L.feat <= 0) return fail(GNNAGG_ERR_ARG, "feature length must be >= 1");
    if (L.tile.on && (L.n1 > 0 || L.accumulate)) return fail(GNNAGG_ERR_STATE, "internal: tiled launch with segments");
    const Geometry g = L.tile.on ? tile_geometry(L.tile, L.feat) : pick_geometry(L.feat, L.x, L.y, L.partial, L.feat);
    const bool is_max = L.reduce == GNNAGG_REDUCE_MAX;
    PlanArgs a;
    a.t0 = reinterpret_cast<const int4 *>(L.t0); a.t1 = reinterpret_cast<const int4 *>(L.t1);
    a.idx = L.idx; a.val = L.val; a.x = L.x; a.y = L.y; a.partial = L.partial;
    a.n0 = L.n0; a.n1 = L.n1; a.feat = L.feat; a.ntiles = g.ntiles; a.chunk = L.chunk;
    a.mean = L.reduce == GNNAGG_REDUCE_MEAN; a.remap = L.xcd_remap; a.accumulate = L.accumulate; a.relu = L.relu;
    a.row_aux = L.row_aux;
    a.slot_hub = L.slot_hub; a.mrow_ptr = L.hubs.mrow_ptr; a.mrow_id = L.hubs.mrow_id; a.row_ptr = L.row_ptr;
    a.hub_count = L.hub_count; a.hub_count_stride = L.hub_count_stride; a.partial_bytes = 0;
    a.xpitch = L.feat; a.ppitch = L.feat; a.x_tile_stride = a.p_tile_stride = g.group * g.vec; a.yvec = g.vec;
    a.tile_major = 0; a.item_blocks = 0; a.ptile_bytes = 0; a.probe_sink = nullptr;
    if (L.tile.on) {
        a.xpitch = L.tile.xpitch; a.x_tile_stride = L.tile.x_tile_stride; a.ppitch = L.tile.ppitch;
        a.p_tile_stride = L.tile.p_tile_stride; a.yvec = L.tile.yvec; a.tile_major = 1;
        const size_t tb = (size_t)L.hubs.n_slots * L.tile.ppitch * sizeof(float);
        a.ptile_bytes = tb < 0x7fffffffULL ? (unsigned)tb : 0u;
    }
    {
        const size_t pbytes = (size_t)L.hubs.n_slots * L.feat * sizeof(float);
        if (L.hubs.n_mrows == 0 || pbytes >= 0x7fffffffULL || g.ntiles > L.hub_count_stride) a.hub_count = nullptr;
        else a.partial_bytes = (unsigned)pbytes;
    }
    const bool hubs_in_kernel = a.hub_count != nullptr;
    {
        const size_t ybytes = (size_t)L.num_rows * L.feat * sizeof(float);
        a.wt = (!L.accumulate && ybytes < 0x7fffffffULL) ? 1 : 0;
        a.ybytes = (unsigned)ybytes;
    }
    // dense combine fused as the epilogue when one lane group spans the row and the [32][K] tile fits LDS
    const bool want_nn = L.nn_weight != nullptr;
    // (8-lane groups, F <= 32: the GEMM is ~11 us on the arxiv-shaped input and the epilogue costs as much -- not fused)
    const bool fuse_nn = want_nn && !L.tile.on && !L.probe && g.ntiles == 1 && g.group >= 16 && !L.accumulate && !L.relu && !L.t0_partials;
    // 4 gathers per batch on the 32-lane float4 geometry when the caller asks for it (see k_gcn_plan)
    const bool u4 = L.unroll == 4 && g.vec == 4 && (g.group == 32 || g.group == 64) && !L.tile.on && !fuse_nn;
    const int blk = block_for(g.group);
    const int gpb = fuse_nn ? std::max(kNnRows, blk / g.group) : blk / g.group;
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
    if (L.probe) {
        if (is_max) return fail(GNNAGG_ERR_ARG, "probe: sum/mean only");
        a.probe_sink = device_probe_sink();
        if (!a.probe_sink) return fail(GNNAGG_ERR_HIP, "probe: no sink");
        if (grid > 0) {
#define CALL_PROBE hipLaunchKernelGGL((k_gcn_plan<VEC, GROUP, false, true>), dim3(grid), dim3(blk), 0, stream, a);
            if (u4 && g.group == 32) hipLaunchKernelGGL((k_gcn_plan<4, 32, false, true, 4>), dim3(grid), dim3(blk), 0, stream, a);
            else if (u4) hipLaunchKernelGGL((k_gcn_plan<4, 64, false, true, 4>), dim3(grid), dim3(blk), 0, stream, a);
            else DISPATCH_GEOM(g, CALL_PROBE)
#undef CALL_PROBE
            HIP_TRY(hipGetLastError());
        }
        return GNNAGG_OK;
    }
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
        if (u4 && g.group == 32) {
            if (is_max) hipLaunchKernelGGL((k_gcn_plan<4, 32, true, false, 4>), dim3(grid), dim3(blk), 0, stream, a);
            else        hipLaunchKernelGGL((k_gcn_plan<4, 32, false, false, 4>), dim3(grid), dim3(blk), 0, stream, a);
        } else if (u4) {
            if (is_max) hipLaunchKernelGGL((k_gcn_plan<4, 64, true, false, 4>), dim3(grid), dim3(blk), 0, stream, a);
            else        hipLaunchKernelGGL((k_gcn_plan<4, 64, false, false, 4>), dim3(grid), dim3(blk), 0, stream, a);
        } else DISPATCH_GEOM(g, CALL_PLAN)
#undef CALL_PLAN
        HIP_TRY(hipGetLastError());
    }
    GcnLaunch C;
    C.wl = L.hubs; C.row_ptr = L.row_ptr; C.y = L.y; C.partial = L.partial; C.feat = L.feat; C.reduce = L.reduce;
    C.accumulate = L.accumulate; C.relu = L.relu; C.row_aux = L.row_aux;
    const int rc = hubs_in_kernel ? GNNAGG_OK : launch_combine_gcn(C, g, is_max, stream, nullptr, nullptr, 0, &L.tile);
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
    a.n1 = L.n1; a.feat = L.feat; a.ntiles = ceil_div(L.feat, 32); a.mean = L.reduce == GNNAGG_REDUCE_MEAN; a.relu = L.relu;
    a.att = L.att; a.heads = L.heads; a.dhead = L.heads > 0 ? L.feat / L.heads : L.feat; a.slope = L.slope;
    const bool is_max = L.reduce == GNNAGG_REDUCE_MAX;
    const bool is_gat = L.att != nullptr;
    if (is_gat && (a.dhead % 32) != 0) return fail(GNNAGG_ERR_ARG, "long-row GAT kernel needs head width % 32 == 0");
    // 64-column tiles for the hub form where the launch is bound by consumer throughput, not by one row's chain (see the kernel)
    const bool wide = !L.medium && !is_gat && L.feat > 32 && (L.tile_w == 64 || (L.tile_w == 0 && (long)a.n1 * a.ntiles > 2L * device_cu_count()));
    if (wide) a.ntiles = ceil_div(L.feat, 64);
    const int grid = a.n1 * a.ntiles;
#define LAUNCH_LONG(V, B)                                                                                            \
    {                                                                                                                \
        const size_t lds = (size_t)long_round_edges<V, B>() * (2 * 32 + 2) * sizeof(float);                          \
        if (is_gat)      LAUNCH_LONG_K((k_gcn_rows_long<V, false, true, B>), B)                                      \
        else if (is_max) LAUNCH_LONG_K((k_gcn_rows_long<V, true, false, B>), B)                                      \
        else             LAUNCH_LONG_K((k_gcn_rows_long<V, false, false, B>), B)                                     \
    }
#define LAUNCH_WIDE(V)                                                                                               \
    {                                                                                                                \
        const size_t lds = (size_t)long_round_edges<V, kLongBlock, 64>() * (2 * 64 + 2) * sizeof(float);             \
        if (is_max) LAUNCH_LONG_K((k_gcn_rows_long<V, true, false, kLongBlock, 64>), kLongBlock)                     \
        else        LAUNCH_LONG_K((k_gcn_rows_long<V, false, false, kLongBlock, 64>), kLongBlock)                    \
    }
#define LAUNCH_LONG_K(K, B)                                                                                          \
    {                                                                                                                \
        static OncePerDevice big_lds_ok; /* > 64 KB of dynamic LDS needs the attribute: once per instantiation AND device */ \
        if (lds > 64 * 1024 && big_lds_ok.first()) {                                                                 \
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&K), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            big_lds_ok.done();                                                                                       \
        }                                                                                                            \
        hipLaunchKernelGGL(K, dim3(grid), dim3(B), lds, stream, a);                                                  \
    }
    if (!L.medium && !is_gat && !wide && vec == 4) {   // hub rows, 16-byte lanes: two consumer wavefronts (see the kernel)
        const size_t lds = (size_t)long_round_edges<4, kLongBlock2, 32, 2>() * (2 * 32 + 2) * sizeof(float) + 64 * sizeof(unsigned long long);
        if (is_max) LAUNCH_LONG_K((k_gcn_rows_long<4, true, false, kLongBlock2, 32, 2>), kLongBlock2)
        else        LAUNCH_LONG_K((k_gcn_rows_long<4, false, false, kLongBlock2, 32, 2>), kLongBlock2)
    } else if (wide) {
        if (vec == 4) LAUNCH_WIDE(4) else if (vec == 2) LAUNCH_WIDE(2) else LAUNCH_WIDE(1)
    } else if (L.medium) {
        if (vec == 4) LAUNCH_LONG(4, kMediumBlock) else if (vec == 2) LAUNCH_LONG(2, kMediumBlock) else LAUNCH_LONG(1, kMediumBlock)
    } else {
        if (vec == 4) LAUNCH_LONG(4, kLongBlock) else if (vec == 2) LAUNCH_LONG(2, kLongBlock) else LAUNCH_LONG(1, kLongBlock)
    }
#undef LAUNCH_LONG_K
#undef LAUNCH_LONG
#undef LAUNCH_WIDE
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
        if (!L.timer) return GNNAGG_OK;  // size query only (pass the run's x / y so the lane geometry is the run's)
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
        // the size query saw no feature pointers: an unaligned x / y picks narrower lanes and a larger grid than it reported
        if (a.timer && grid > L.timer_capacity)
            return fail(GNNAGG_ERR_ARG, "run_clock: timer buffer too small for this launch (query num_blocks with the same x / y pointers)");
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

}  // namespace gnnagg
