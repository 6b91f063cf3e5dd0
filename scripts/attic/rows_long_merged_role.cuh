// NOT PART OF THE BUILD -- kept as the record of an experiment (round 3).
// The long rows of the rows mode as a ROLE inside the short-row launch (256-thread workgroups, 17 KB of static LDS, gather groups
// taking turns in three classes) instead of a kernel of their own on an auxiliary stream.  Bit-exact (91 parity tests green), and
// a balanced launch after a rows launch kept its stand-alone time (74.0 vs 74.5 us) -- but the merged kernel needs the hub role's
// 160 VGPRs for every workgroup, so the short rows run at 3 instead of 6 wavefronts per SIMD, and the hub role has 384 instead of
// 896 edges in flight: arxiv-shaped rows mode 182 -> 228 us, products-shaped 9.7 -> 15.2 ms (profiles/r03/rows_mode_merged_role.txt).
// hipExtAnyOrderLaunch (two kernels overlapping on ONE stream) is documented as unsupported on gfx9 boards (hip_ext.h:67).  The
// two-stream form of round 2 stays; the reference-facing surfaces no longer enter the rows mode by default ("reference_defaults").
#pragma once
// rows_long.cuh -- the long rows of the canonical rows mode (`scheduled = 0`: one sequential FMA chain per (row, column) in CSR
// order, aggr_gcn.h:13-35 / aggr_gat.h:125-163), as a ROLE inside the short-row launch (k_gcn_rows, k_gat_rows).
//
// Round 2 ran these rows in a kernel of their own (512 threads, 115 KB of LDS per workgroup) on an auxiliary stream beside the
// short rows.  A second HIP stream costs every later launch of the process ~4 us ("Platform finding"), and the fork /
// join itself ~23 us, so the long rows are now workgroups at the head of the short-row kernel's grid: 256 threads, 17 KB of LDS
// (static: 9 workgroups per CU still fit, the short rows keep their occupancy), one launch, one stream.
//
// One workgroup per (row, 32-column tile).  The chain cannot be split, the GATHERS can: wavefronts 1-3 (192 threads = 24 / 12 /
// 6 gather groups of 8 / 16 / 32 lanes for float4 / float2 / float rows) fetch the 128-byte tile segments of different neighbors in
// parallel, U = 8 per group, TWO register sets per group -- 384 edges in flight per workgroup -- and wavefront 0 only consumes:
// lane c owns column c and runs its chain from LDS, 4 chain steps per ds_read_b128.  What keeps the LDS small is that the
// in-flight window is decoupled from the staged round: a round is what ONE THIRD of the groups hold (64 edges for float4 rows);
// the three classes of groups take turns writing a round into one of two stage buffers while the consumer reads the other, one
// barrier per round.  Stage layout as before: the values of 4 consecutive edges of one column are contiguous, XOR-swizzled so a
// store instruction covers all banks.
#include "kernel_util.cuh"

namespace gnnagg {

struct RowsLongArgs {
    const int4 *r1;  // {beg, end, row, -} per long row, heaviest first
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

static constexpr int kLongBlock = 256;                      // threads of a long-row workgroup = block size of the merged kernels
static constexpr int kLongGatherThreads = kLongBlock - 64;  // wavefront 0 only consumes
static constexpr int kLongU = 8;                            // neighbors per gather group per round
static constexpr int kLongClasses = 3;                      // gather groups take turns: class (round % 3) stages the round

template <int VEC>
constexpr int long_round_edges() { return (kLongGatherThreads / (32 / VEC)) / kLongClasses * kLongU; }
// LDS floats of a long-row workgroup: two stage buffers [RE][32] + two weight buffers [RE]
template <int VEC>
constexpr int long_lds_floats() { return long_round_edges<VEC>() * (2 * 32 + 2); }

template <int VEC, bool IS_MAX, bool IS_GAT>
__device__ __forceinline__ void rows_long_body(const RowsLongArgs &a, int job, float *lds)
{
    constexpr int GL = 32 / VEC;                 // lanes of one gather group: GL * VEC = 32 columns = 128 bytes
    constexpr int NG = kLongGatherThreads / GL;  // gather groups per workgroup
    constexpr int GPR = NG / kLongClasses;       // groups that stage one round
    constexpr int U = kLongU;
    constexpr int RE = GPR * U;                  // edges per round
    static_assert(NG % kLongClasses == 0 && RE % 4 == 0, "gather groups must split into classes of whole quads");
    // element (edge k of the round, column c) at ((k/4) * 32 + (c ^ swz(k/4))) * 4 + k % 4, swz(q) = (q >> 1) & 3
    float *stage0 = lds, *stage1 = lds + RE * 32, *wst0 = lds + 2 * RE * 32, *wst1 = wst0 + RE;
    const int F = a.feat;
    const int tile = job % a.ntiles32;
    const int4 d = a.r1[job / a.ntiles32];
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
            // 16 chain steps per batch = 4 + 4 ds_read_b128, the next batch's reads issued before the current batch's steps
            auto load16 = [&](float4 (&xs)[4], float4 (&ws)[4], int k) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int kq = (k >> 2) + q;
                    xs[q] = *reinterpret_cast<const float4 *>(&stage[(kq * 32 + (c ^ ((kq >> 1) & 3))) * 4]);
                    ws[q] = *reinterpret_cast<const float4 *>(&wst[k + 4 * q]);
                }
            };
            auto steps16 = [&](const float4 (&xs)[4], const float4 (&ws)[4]) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    step(xs[q].x, ws[q].x);
                    step(xs[q].y, ws[q].y);
                    step(xs[q].z, ws[q].z);
                    step(xs[q].w, ws[q].w);
                }
            };
            const int nfull = n & ~15;
            int k0 = 0;
            if (nfull > 0) {
                float4 xa[4], wa[4], xb[4], wb[4];
                load16(xa, wa, 0);
                while (true) {
                    if (k0 + 16 < nfull) load16(xb, wb, k0 + 16);
                    steps16(xa, wa);
                    k0 += 16;
                    if (k0 >= nfull) break;
                    if (k0 + 16 < nfull) load16(xa, wa, k0 + 16);
                    steps16(xb, wb);
                    k0 += 16;
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
    // ---- gather wavefronts.  Group g = class cg, slot sg: it fetches edges [r * RE + sg * U, + U) of the rounds r = cg, cg + 3, ...
    const int t = (int)threadIdx.x - 64;
    const int g = t / GL, lane = t & (GL - 1);
    const int cg = g / GPR, sg = g - cg * GPR;
    const int col = tile * 32 + lane * VEC;
    const bool col_ok = col < F;
    const float *__restrict__ xcol = a.x + col;
    const float a_dst = IS_GAT ? a.att[((size_t)d.z * a.heads + head) * 2] : 0.0f;
    // Two of the group's rounds are in flight in registers (sets A / B), and the neighbor ids / weights are fetched two of ITS
    // rounds before their gathers.  Every lane of a group loads the group's U ids (same addresses: one request each); lane u < U
    // also carries the u-th edge's weight and writes it to LDS.  Branch-free: edges past the row's end are clamped to the last
    // edge (their stage slots are never read).
    const int mlane = lane < U ? lane : U - 1;
    struct Meta {
        int sid[U];
        float w;
    };
    auto meta_load = [&](int k, Meta &m) {   // the group's k-th round
        const long e0 = (long)d.x + (long)(kLongClasses * k + cg) * RE + sg * U;
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
                const int kq = (sg * U + 4 * hq) >> 2;
                const int swz = (kq >> 1) & 3;
#pragma unroll
                for (int j = 0; j < VEC; ++j)
                    *reinterpret_cast<float4 *>(&stage[(kq * 32 + ((lane * VEC + j) ^ swz)) * 4]) =
                        make_float4(xv[4 * hq].v[j], xv[4 * hq + 1].v[j], xv[4 * hq + 2].v[j], xv[4 * hq + 3].v[j]);
            }
        }
        if (lane < U) wst[sg * U + lane] = IS_GAT ? edge_weight(a_dst, wv, a.slope) : wv;
    };
    Pack<VEC> xa[U], xb[U];
    float wa = 0.0f, wb = 0.0f;
    Meta ma, mb;  // metadata of the next issue of set A / set B
    meta_load(0, ma);
    meta_load(1, mb);
    issue(ma, xa, wa);
    meta_load(2, ma);
    issue(mb, xb, wb);
    meta_load(3, mb);
    for (int r = 0; r < nrounds; ++r) {
        if (r % kLongClasses == cg) {
            const int k = r / kLongClasses;   // this group's k-th round: set A for even k, set B for odd k
            float *stage = (r & 1) ? stage1 : stage0, *wst = (r & 1) ? wst1 : wst0;
            if ((k & 1) == 0) {
                stage_round(xa, wa, stage, wst);
                if (r + 2 * kLongClasses < nrounds) { issue(ma, xa, wa); meta_load(k + 4, ma); }
            } else {
                stage_round(xb, wb, stage, wst);
                if (r + 2 * kLongClasses < nrounds) { issue(mb, xb, wb); meta_load(k + 4, mb); }
            }
        }
        __syncthreads();  // round r is staged (and the consumer is done with round r - 1's buffer, which round r + 1 overwrites)
    }
}

}  // namespace gnnagg
