#pragma once
// combine.cuh -- ordered combine of the partial rows of split rows (shared by the GCN and the GAT translation units).
#include "kernel_util.cuh"

namespace gnnagg {

struct CombineArgs {
    const int *mrow_id, *mrow_ptr, *row_ptr;
    const int *big_rows;  // indices into mrow_* of the rows with more than kBigRowPartials partials
    int n_big, nblocks_small;
    int accumulate;  // 1: y += (sum of partials)
    int relu = 0;    // 1: y = max(result, 0) (GCN)
    const int *row_aux = nullptr;  // gnnagg_set_row_aux (finish_gcn_row)
    // run_with_nn: nn_out[row, :] = (finished row) . nn_weight for the rows finished here (ntiles == 1)
    const float *nn_weight;
    float *nn_out;
    int nn_cols;
    const float *partial;
    const float *partial_den;  // GAT only
    float *y;
    int n_mrows, feat, ntiles, heads, dhead, mean;
    // addressing of the partial scratch (floats): row-major (ppitch = feat, p_tile_stride = the lane group's column span) or
    // the 2-D blocked mode's tile-major image; yvec = alignment class of the Y rows (below VEC: element-wise stores)
    int ppitch, yvec;
    long p_tile_stride;
};

static inline void combine_strides(CombineArgs &c, int feat, const Geometry &g, const TileSpec *tile)
{
    c.ppitch = feat; c.p_tile_stride = g.group * g.vec; c.yvec = g.vec;
    if (tile && tile->on) { c.ppitch = tile->ppitch; c.p_tile_stride = tile->p_tile_stride; c.yvec = tile->yvec; }
}

// Adds the partial rows of every split row in ascending slot order (deterministic counterpart of
// the reference's atomicAdd, aggr_gcn.h:112) and applies mean / softmax normalisation.
static constexpr int kCombineBatch = 16;   // partial rows a lane group keeps in flight
static constexpr int kCombineStage = 128;  // partial rows a workgroup stages in LDS per round (big rows)

// BIG = true: the workgroup-per-row path only (64 KB of LDS staging), launched over the rows of big_rows; BIG = false: the
// lane-group path only, with 4 KB of LDS (the two used to be one kernel, and the staging array capped the lane-group path
// at two workgroups per CU -- it now runs for every row of the source-partitioned mode).
template <int VEC, int GROUP, bool IS_MAX, bool IS_GAT, bool BIG>
__global__ __launch_bounds__(kBlock) void k_combine(const CombineArgs a)
{
    constexpr int ITEMS = kBlock / GROUP;
    const int F = a.feat;
    __shared__ float stage[BIG ? kCombineStage * GROUP * VEC : kBlock * VEC];
    __shared__ float stage_den[(BIG && IS_GAT) ? kCombineStage * 64 : 1];
    const bool nn = !IS_GAT && a.nn_weight != nullptr;
    if constexpr (BIG) {
        // ---- big rows (hubs: hundreds of partials): one workgroup per (row, column tile).  All lane
        // groups fetch partial rows in parallel into LDS (kCombineStage rows per round, kCombineBatch
        // loads in flight per group), then each column is summed from LDS in ascending slot order.
        const int bb = (int)blockIdx.x;
        const int tile = bb % a.ntiles;
        const int m = a.big_rows[bb / a.ntiles];
        const int s0 = a.mrow_ptr[m], s1 = a.mrow_ptr[m + 1];
        const int row = a.mrow_id[m];
        const int grp = (int)threadIdx.x / GROUP, lane = threadIdx.x & (GROUP - 1);
        const int col0 = tile * GROUP * VEC;
        const int col = col0 + lane * VEC;
        constexpr int W = GROUP * VEC;                // columns of this tile
        const float *__restrict__ ptile = a.partial + (size_t)tile * a.p_tile_stride;
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
                    if (p0 + u < nst && col < F) p[u] = load_pack<VEC>(ptile + (size_t)(sb + p0 + u) * a.ppitch + lane * VEC);
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
                if (a.accumulate) acc = a.y[(size_t)row * F + col0 + c] + acc;
            } else {
                float one[1] = {acc};
                finish_gcn_row<1, IS_MAX>(one, a.mean ? a.row_ptr[row + 1] - a.row_ptr[row] : 1, row, a.y + (size_t)row * F + col0 + c, a.mean,
                                          a.accumulate, a.relu, a.row_aux);
                acc = one[0];
            }
            a.y[(size_t)row * F + col0 + c] = acc;
            if (nn) stage[c] = acc;  // the staging rounds are over
        }
        if (!nn) return;
        __syncthreads();
        row_times_weight(stage, F, a.nn_weight, a.nn_cols, a.nn_out + (size_t)row * a.nn_cols, (int)threadIdx.x, kBlock);
        return;
    } else {
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
        if (a.n_big > 0 && s1 - s0 > kBigRowPartials) here = false;  // handled by the workgroup-per-row path
    }
    const bool active = here && col < a.feat;
    if (!nn && !active) return;
    const float *__restrict__ ptile = a.partial + (size_t)tile * a.p_tile_stride;
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
                    p[u] = load_pack<VEC>(ptile + (size_t)(sb + u) * a.ppitch + lane * VEC);
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
            if (a.accumulate) {
                const Pack<VEC> old = load_pack<VEC>(a.y + (size_t)row * F + col);
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[k] = old.v[k] + acc[k];
            }
        } else {
            finish_gcn_row<VEC, IS_MAX>(acc, a.mean ? a.row_ptr[row + 1] - a.row_ptr[row] : 1, row, a.y + (size_t)row * F + col, a.mean,
                                        a.accumulate, a.relu, a.row_aux);
        }
        if (a.yvec < VEC || F - col < VEC) store_pack_any<VEC>(a.y + (size_t)row * F + col, acc, F - col, a.yvec);
        else store_pack<VEC>(a.y + (size_t)row * F + col, acc);
        if (nn) store_pack<VEC>(&stage[grp * GROUP * VEC + col], acc);  // ntiles == 1: col = lane * VEC
    }
    if (!nn) return;
    __syncthreads();
    if (here) row_times_weight(&stage[grp * GROUP * VEC], F, a.nn_weight, a.nn_cols, a.nn_out + (size_t)row * a.nn_cols, lane, GROUP);
    }
}

}  // namespace gnnagg
