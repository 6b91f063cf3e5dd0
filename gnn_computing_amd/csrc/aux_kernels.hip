// aux_kernels.hip -- everything beside the two aggregation families: CSR -> edge list, the unfused edge-softmax stages,
// the edge-wise and naive SpMM baselines, validators, the dense combine GEMM, the CSR check and the halo pack.
#include "kernel_util.cuh"

#include <type_traits>

namespace gnnagg {

// ------------------------------------------------------------------------- CSR -> edge list
// A lane group strides over the edges of one row; group size follows the average degree so short
// rows do not idle a whole wavefront.
static int edge_group(int avg_deg)
{
    int g = 8;
    while (g < 64 && g < avg_deg) g <<= 1;
    return g;
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
    // ascending order kept; the loads of 16 partials are issued together (a hub has hundreds: one dependent load per
    // step made this 16 us on the arxiv-shaped input)
    float s = 0.0f;
    const int p1 = mrow_ptr[m + 1];
    for (int p = mrow_ptr[m]; p < p1; p += 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = p + u < p1 ? partial_den[(size_t)(p + u) * H + h] : 0.0f;
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (p + u < p1) s += v[u];
    }
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
    { const int rcz = launch_zero_words(y, (size_t)V * feat, stream); if (rcz) return rcz; }  // aggr_gcn.h:448
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

// The narrow layers (K = 32 / 64 / 96 / 128 -> 32 or 64 columns) are one round of workgroups: 1323 tiles of 128 rows all resident at once,
// and in k_dense_nn each of them walks its K chunks one after the other -- request, wait, LDS, barrier, 16 MFMAs, barrier -- so a tile is
// NCH HBM latencies in a row and the launch is as long as that chain (27 us for 169 343 x 128 @ 128 x 32, half the HBM roofline, the same as
// rocBLAS).  Here ALL of a tile's chunks are requested before anything else happens, in straight-line code (NCH is a template parameter: no
// loop back-edge for the compiler's vmcnt bookkeeping to get lost at), and they are written to LDS and multiplied in order while the later
// ones are still in flight; the tile's whole B strip (K x 32 floats) goes to LDS once.  Same operand layout, same ascending-k chain per
// output as k_dense_nn: bit-exact.  16-byte aligned A rows, K = 32 NCH <= 128.
template <int NCH, int NCB>
__global__ __launch_bounds__(256) void k_dense_nn_up(const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ C, int M, int N)
{
    // NCB = 32-column blocks per workgroup (1: N <= 32; 2: N <= 64 -- the tile of A is read once for both, not once per column block)
    constexpr int K = NCH * kGemmKC, BC = NCB * kGemmCols, NIMG = NCB == 1 ? 2 : 1;   // (two A images only where the LDS has room for them)
    __shared__ float As[NIMG][kGemmRows * kGemmPitch];
    __shared__ float Bs[K * BC];
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const int row0 = blockIdx.x * kGemmRows, col0 = blockIdx.y * BC;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // A through a descriptor rebased to the tile: rows beyond M are outside it (zeros)
    const size_t a_off = (size_t)row0 * K * sizeof(float), a_all = (size_t)M * K * sizeof(float);
    const size_t a_left = a_off < a_all ? a_all - a_off : 0;
    const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(A) + (size_t)row0 * K, 0,
                                                                            (int)(unsigned)(a_left < 0xfffffffcULL ? a_left : 0xfffffffcULL), 0x00020000);
    u4 ra[NCH][4];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            ra[c][j] = __builtin_amdgcn_raw_buffer_load_b128(arsrc, ((((int)threadIdx.x >> 3) + 32 * j) * K + c * kGemmKC + ((int)threadIdx.x & 7) * 4) * (int)sizeof(float), 0, 0);
    // the B strip: K x BC floats, NCH x NCB x 4 per thread (column guard by clamp + select: no branch)
    constexpr int NBV = NCH * NCB * 4;
    float rb[NBV];
#pragma unroll
    for (int j = 0; j < NBV; ++j) {
        const int e = (int)threadIdx.x + 256 * j, kk = e / BC, cc = col0 + (e % BC);
        const float v = B[(size_t)kk * N + (cc < N ? cc : N - 1)];
        rb[j] = cc < N ? v : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < NBV; ++j) Bs[(int)threadIdx.x + 256 * j] = rb[j];
    f32x16 acc[NCB];
#pragma unroll
    for (int n = 0; n < NCB; ++n)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[n][i] = 0.0f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        float *as = As[NIMG == 2 ? (c & 1) : 0];
        if (NIMG == 1 && c > 0) __syncthreads();   // one image: every wavefront is done with chunk c - 1 before chunk c overwrites it
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float *d = as + (((int)threadIdx.x >> 3) + 32 * j) * kGemmPitch + ((int)threadIdx.x & 7) * 4;
            d[0] = __uint_as_float(ra[c][j][0]); d[1] = __uint_as_float(ra[c][j][1]); d[2] = __uint_as_float(ra[c][j][2]); d[3] = __uint_as_float(ra[c][j][3]);
        }
        __syncthreads();   // (two images: chunk c + 1 is written while chunk c is still being read by slower wavefronts)
        float av[kGemmKC / 2];
#pragma unroll
        for (int t = 0; t < kGemmKC / 2; ++t) av[t] = as[(wave * 32 + (lane & 31)) * kGemmPitch + 2 * t + (lane >> 5)];
#pragma unroll
        for (int n = 0; n < NCB; ++n) {
            float bv[kGemmKC / 2];
#pragma unroll
            for (int t = 0; t < kGemmKC / 2; ++t) bv[t] = Bs[(c * kGemmKC + 2 * t + (lane >> 5)) * BC + n * kGemmCols + (lane & 31)];
#pragma unroll
            for (int t = 0; t < kGemmKC / 2; ++t) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[t], acc[n], 0, 0, 0);
        }
    }
    // C/D layout of 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int n = 0; n < NCB; ++n) {
        const int col = col0 + n * kGemmCols + (lane & 31);
        if (col < N) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = row0 + wave * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                if (row < M) C[(size_t)row * N + col] = acc[n][reg];
            }
        }
    }
}

// Wide-N variant (N > 64: the 512 -> 128 layer of the 3-layer models): a workgroup owns a TM x 128 output tile (TM = 128 for the
// bulk), so A is read from HBM once whatever N is (k_dense_nn re-reads it per 32-column block); wavefront w owns the tile's columns
// [32 w, 32 w + 32) over all TM rows -- TM / 32 accumulators of 32 x 32: a k-step of 2 feeds TM / 32 MFMAs from TM / 32 + 1 operand
// reads (k_dense_nn: 1 MFMA from 2), which is what the matrix pipe needs to stay busy (k_dense_nn: 59 TFLOP/s on 169 343 x 512 @
// 512 x 128, rocBLAS 89).  K in chunks of 32 through a double-buffered LDS image (A: pitch 33, B: as is): the next chunk's global
// loads are in flight during the current chunk's MFMAs, one barrier per chunk.  Every accumulator is still one ascending-k chain of
// v_mfma_f32_32x32x2_f32 steps: bit-exact against the oracle's GEMM like the other two kernels.
// Tail: two workgroups fit a CU (LDS, 256 registers), so the chip runs 2 x CUs tiles at a time and 169 343 rows = 1323 tiles of 128
// are 2.58 rounds of 512 -- the third round 58 % full.  The rows beyond the last full round are cut into at most one round of
// SMALLER tiles instead (TMT = 32 / 64 / 96 rows: 2 rounds of 128 + 1 of 96 here, 2.75 round-times instead of 3).
// (K chunk 16 with 3 or 4 workgroups per CU, 8 with 4: 273-294 us on 169 343 x 512 @ 512 x 128 against 266 us for 32 with 2 --
// more wavefronts per SIMD buy nothing here)
#ifndef GNNAGG_GEMM_KC
#define GNNAGG_GEMM_KC 32
#endif
#ifndef GNNAGG_GEMM_AHEAD_PIPE
#define GNNAGG_GEMM_AHEAD_PIPE 1   // k_dense_nn_ahead: operand reads one k-pair ahead of the MFMAs (0: the compiler's order; 2: the reads spread between the MFMAs)
#endif
#ifndef GNNAGG_GEMM_AHEAD
#define GNNAGG_GEMM_AHEAD 1   // 1: 16-byte-aligned lean shapes with K % 32 == 0 run on k_dense_nn_ahead (chunks requested two periods ahead, hand-counted vmcnt); 0: k_dense_nn_lean<4>
#endif
#ifndef GNNAGG_GEMM_SWAP
#define GNNAGG_GEMM_SWAP 0   // lean form, A/B switch: 1 = operands swapped, transposed accumulator, 16-byte C stores (measured equal: profiles/r05/gemm_forms.txt)
#endif
static constexpr int kBigT = 128, kBigKC = GNNAGG_GEMM_KC, kBigPA = kBigKC + 1;
#ifndef GNNAGG_GEMM_WGS
#define GNNAGG_GEMM_WGS (GNNAGG_GEMM_KC <= 16 ? 3 : 2)
#endif
static constexpr int kBigWgs = GNNAGG_GEMM_WGS;   // workgroups per CU (LDS: 33.8 KB each at 16, 66.6 KB at 32)

// AV: floats per aligned load of A (4: K % 4 == 0 and A 16-byte aligned; 2: K even, A 8-byte aligned -- the 602-wide layer; 1: any).
// AV > 1 also says N % 4 == 0 and B 16-byte aligned: every 4-float piece of B is one aligned load.
// (Round 3's k_dense_nn_big -- one 128-row tile per workgroup, a last round of smaller tiles -- is gone: the strips below do the same
// arithmetic without a partial last round; its text is in git history, its numbers in profiles/r03/gemm.txt.)
#ifdef GNNAGG_GEMM_TIMELINE   // A/B builds only (scripts/history/exp_gemm_timeline.py): s_memtime stamps of wave 0 of every workgroup, six per chunk
__device__ unsigned long long *g_gemm_tl = nullptr;
#define TL_STAMP(slot) do { if (g_gemm_tl && threadIdx.x == 0 && g < 64) g_gemm_tl[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 64 + g) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TL_STAMP(slot) do { } while (0)
#endif
template <int AV>
__global__ __launch_bounds__(256, kBigWgs) void k_dense_nn_strip(const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ C,
                                                                 int M, int N, int K, int nb32, int nstrips)
{
    extern __shared__ float lds[];
    constexpr int KQ = kBigKC / 4, NA = kBigT * KQ / 256, NB = kBigKC * (kBigT / 4) / 256, BV = AV > 1 ? 4 : 1;
    const int col0 = blockIdx.y * kBigT;
    const int q = nb32 / nstrips, extra = nb32 - q * nstrips, sidx = blockIdx.x;
    const int blk0 = sidx * q + (sidx < extra ? sidx : extra), nblk = q + (sidx < extra ? 1 : 0);
    if (nblk == 0) return;
    float *As0 = lds, *As1 = lds + kBigT * kBigPA, *Bs0 = lds + 2 * kBigT * kBigPA, *Bs1 = Bs0 + kBigKC * kBigT;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    float4 ra[NA], rb[NB];
    auto load_a = [&](int r, int c) -> float4 {   // clamped (always valid) addresses; zeroing happens at the stash
        const int rc = r < M ? r : M - 1;
        const float *src = A + (size_t)rc * K;
        if constexpr (AV == 4) return *reinterpret_cast<const float4 *>(src + (c < K ? c : K - 4));
        else if constexpr (AV == 2) {
            const float2 lo = *reinterpret_cast<const float2 *>(src + (c < K ? c : K - 2)), hi = *reinterpret_cast<const float2 *>(src + (c + 2 < K ? c + 2 : K - 2));
            return make_float4(lo.x, lo.y, hi.x, hi.y);
        } else return make_float4(src[c < K ? c : K - 1], src[c + 1 < K ? c + 1 : K - 1], src[c + 2 < K ? c + 2 : K - 1], src[c + 3 < K ? c + 3 : K - 1]);
    };
    auto load_b = [&](int r, int c) -> float4 {
        const int rc = r < K ? r : K - 1;
        const float *src = B + (size_t)rc * N;
        if constexpr (BV == 4) return *reinterpret_cast<const float4 *>(src + (c < N ? c : N - 4));
        else return make_float4(src[c < N ? c : N - 1], src[c + 1 < N ? c + 1 : N - 1], src[c + 2 < N ? c + 2 : N - 1], src[c + 3 < N ? c + 3 : N - 1]);
    };
    auto keep4 = [](float4 v, bool k0, bool k1, bool k2, bool k3) -> float4 {   // bit masks, not selects
        v.x = __uint_as_float(__float_as_uint(v.x) & (k0 ? 0xffffffffu : 0u)); v.y = __uint_as_float(__float_as_uint(v.y) & (k1 ? 0xffffffffu : 0u));
        v.z = __uint_as_float(__float_as_uint(v.z) & (k2 ? 0xffffffffu : 0u)); v.w = __uint_as_float(__float_as_uint(v.w) & (k3 ? 0xffffffffu : 0u));
        return v;
    };
    auto fetch = [&](int row0, int k0) {
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int p = (int)threadIdx.x + 256 * j;
            ra[j] = load_a(row0 + p / KQ, k0 + (p % KQ) * 4);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) rb[j] = load_b(k0 + (threadIdx.x >> 5) + 8 * j, col0 + (threadIdx.x & 31) * 4);
    };
    auto stash = [&](float *As, float *Bs, int row0, int k0) {
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int p = (int)threadIdx.x + 256 * j;
            const int r = row0 + p / KQ, c = k0 + (p % KQ) * 4;
            const bool rok = r < M;
            const float4 v = AV == 4 ? keep4(ra[j], rok && c < K, rok && c < K, rok && c < K, rok && c < K)
                             : AV == 2 ? keep4(ra[j], rok && c < K, rok && c < K, rok && c + 2 < K, rok && c + 2 < K)
                                       : keep4(ra[j], rok && c < K, rok && c + 1 < K, rok && c + 2 < K, rok && c + 3 < K);
            float *da = As + (p / KQ) * kBigPA + (p % KQ) * 4;
            da[0] = v.x; da[1] = v.y; da[2] = v.z; da[3] = v.w;
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int r = k0 + (threadIdx.x >> 5) + 8 * j, c = col0 + (threadIdx.x & 31) * 4;
            const bool rok = r < K;
            *reinterpret_cast<float4 *>(Bs + ((threadIdx.x >> 5) + 8 * j) * kBigT + (threadIdx.x & 31) * 4) =
                BV == 4 ? keep4(rb[j], rok && c < N, rok && c < N, rok && c < N, rok && c < N)
                        : keep4(rb[j], rok && c < N, rok && c + 1 < N, rok && c + 2 < N, rok && c + 3 < N);
        }
    };
    // rbk: 32-row blocks of the tile (workgroup-uniform): the blocks beyond it are skipped by scalar branches (one code path: four
    // unrolled variants of the chunk keep four operand sets alive and spill)
    auto mma = [&](int rbk, const float *As, const float *Bs) {
        const float *ap = As + (lane & 31) * kBigPA + (lane >> 5);
        const float *bp = Bs + (lane >> 5) * kBigT + 32 * wave + (lane & 31);
        // operands of k-step t + 1 are requested before the MFMAs of k-step t (4 x 64 pipe cycles cover the LDS latency); rows of the
        // image that belong to no block of the tile are read and never used
        float a_cur[4], a_nxt[4], b_cur, b_nxt;
#pragma unroll
        for (int i = 0; i < 4; ++i) a_cur[i] = ap[i * 32 * kBigPA];
        b_cur = bp[0];
        // (s_setprio around the burst -- so that the two wavefronts that share a SIMD's matrix pipe fall out of phase -- measured:
        // 201.1 against 202.2 us, nothing)
#pragma unroll
        for (int t = 0; t < kBigKC / 2; ++t) {
            if (t + 1 < kBigKC / 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) a_nxt[i] = ap[i * 32 * kBigPA + 2 * (t + 1)];
                b_nxt = bp[2 * (t + 1) * kBigT];
            }
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[0], b_cur, acc[0], 0, 0, 0);
            if (rbk > 1) acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[1], b_cur, acc[1], 0, 0, 0);
            if (rbk > 2) acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[2], b_cur, acc[2], 0, 0, 0);
            if (rbk > 3) acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[3], b_cur, acc[3], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) a_cur[i] = a_nxt[i];
            b_cur = b_nxt;
        }
    };
    const int nchunks = (K + kBigKC - 1) / kBigKC;
    const int ntiles = (nblk + 3) >> 2, total = ntiles * nchunks;   // the strip as ONE sequence of chunks g = tile * nchunks + c
    const int col = col0 + 32 * wave + (lane & 31);
    auto row_of = [&](int g) { return (blk0 + 4 * (g / nchunks)) * 32; };
    auto k_of = [&](int g) { return (g % nchunks) * kBigKC; };
    // While chunk g is multiplied out of LDS buffer g & 1, chunk g + 1 travels to registers; after the MFMAs it goes to the other LDS
    // buffer.  (A second register set -- chunk g + 2 requested during chunk g -- was built and measured: 206.7 against 201-205 us on the
    // 512 -> 128 layer; the loads are not what the matrix pipe waits for.  profiles/r04/gemm.txt)
    fetch(row_of(0), 0);
    stash(As0, Bs0, row_of(0), 0);
    __syncthreads();
    for (int g = 0; g < total; ++g) {
        const float *As = (g & 1) ? As1 : As0, *Bs = (g & 1) ? Bs1 : Bs0;
        const int tile = g / nchunks, c = g - tile * nchunks;
        const int row0 = (blk0 + 4 * tile) * 32;
        const int rbk = nblk - 4 * tile < 4 ? nblk - 4 * tile : 4;   // 32-row blocks of this tile (workgroup-uniform)
        // (unconditional: the last step re-requests the last chunk -- behind a branch the fetch registers meet in phi copies, and the
        // copies wait for the loads that were just issued)
        const int gn = g + 1 < total ? g + 1 : total - 1;
        TL_STAMP(0);
        fetch(row_of(gn), k_of(gn));
        TL_STAMP(1);
        mma(rbk, As, Bs);
        TL_STAMP(2);
        if (c + 1 == nchunks) {
            // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).  Buffer stores: ONE
            // lane offset + a scalar offset per store (64 global addresses computed up front cost 128 registers while the next
            // tile's fetch is in flight), and rows beyond M fall off the end of the buffer -- the hardware drops them
            const size_t co = (size_t)row0 * N * sizeof(float), c_bytes = (size_t)M * N * sizeof(float);   // descriptor rebased to C[row0][0]
            const size_t cr = co < c_bytes ? c_bytes - co : 0;
            const __amdgpu_buffer_rsrc_t crsrc = __builtin_amdgcn_make_buffer_rsrc(C + (size_t)row0 * N, 0, (int)(unsigned)(cr < 0xfffffffcULL ? cr : 0xfffffffcULL), 0x00020000);
            const unsigned voff = (unsigned)(((size_t)(4 * (lane >> 5)) * N + col) * sizeof(float));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (i < rbk && col < N) {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg)
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[i][reg]), crsrc, voff,
                                                              (unsigned)((32 * i + (reg & 3) + 8 * (reg >> 2)) * N) * (unsigned)sizeof(float), 0);
                }
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) acc[i][reg] = 0.0f;
            }
        }
        TL_STAMP(3);
        if (g + 1 < total) stash((g & 1) ? As0 : As1, (g & 1) ? Bs0 : Bs1, row_of(g + 1), k_of(g + 1));   // the other buffer: last read one chunk ago
        TL_STAMP(4);
        __syncthreads();
        TL_STAMP(5);
    }
}

// Round 4, second step: the LEAN form of the strip kernel.  A per-phase timeline (scripts/history/exp_gemm_timeline.py, s_memtime stamps) showed
// where a chunk goes: issuing the 8 loads of the next chunk 20 % of the period, the 64 MFMAs 35 %, the stash 20 %, the barrier 12 % -- and the
// MFMA phase runs at 80 ticks per MFMA, i.e. the two wavefronts of a SIMD ALTERNATE: while one bursts, every other instruction of its
// neighbour (address arithmetic, clamps, masks, LDS writes: ~210 per chunk) trickles out between MFMAs at ~30 ticks apiece.  The matrix
// pipe is busy 2 x 35 %.  So everything that is not an MFMA or an operand read is made cheap in INSTRUCTIONS:
//   * A and B come through buffer descriptors REBASED per chunk in scalar registers (base = A + row0 * K + k0): the per-thread offsets are
//     computed once per kernel, a fetch is 8 buffer loads and a handful of scalar instructions, and rows beyond M / k beyond K are
//     out of range of the descriptor -- the hardware returns zeros, no clamps, no masks;
//   * only the ragged last K chunk (K % 32 != 0: the 602-wide layer) masks, behind a workgroup-uniform branch;
//   * the chunk / tile counters advance by addition (g / nchunks was an integer division per chunk).
// Needs N % 128 == 0, K even and 8-byte aligned rows; other shapes keep k_dense_nn_strip.  Same arithmetic: bit-exact.
template <int AV>
__global__ __launch_bounds__(256, kBigWgs) void k_dense_nn_lean(const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ C,
                                                                int M, int N, int K, int nb32, int nstrips)
{
    static_assert(AV == 4 || AV == 2, "lean form: 16- or 8-byte loads of A");
    extern __shared__ float lds[];
    constexpr int KQ = kBigKC / 4, NA = kBigT * KQ / 256, NB = kBigKC * (kBigT / 4) / 256;
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const int col0 = blockIdx.y * kBigT;
    const int q = nb32 / nstrips, extra = nb32 - q * nstrips, sidx = blockIdx.x;
    const int blk0 = sidx * q + (sidx < extra ? sidx : extra), nblk = q + (sidx < extra ? 1 : 0);
    if (nblk == 0) return;
    float *As0 = lds, *As1 = lds + kBigT * kBigPA, *Bs0 = lds + 2 * kBigT * kBigPA, *Bs1 = Bs0 + kBigKC * kBigT;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    // per-thread constants: byte offsets of this thread's pieces inside a chunk (global) and inside the LDS images
    int voa[NA], vob[NB], la[NA], lb[NB];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int p = (int)threadIdx.x + 256 * j;
        voa[j] = ((p / KQ) * K + (p % KQ) * 4) * (int)sizeof(float);
        la[j] = (p / KQ) * kBigPA + (p % KQ) * 4;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        vob[j] = (((int)(threadIdx.x >> 5) + 8 * j) * N + (int)(threadIdx.x & 31) * 4) * (int)sizeof(float);
        lb[j] = ((int)(threadIdx.x >> 5) + 8 * j) * kBigT + (int)(threadIdx.x & 31) * 4;
    }
    const size_t a_bytes = (size_t)M * K * sizeof(float), b_bytes = (size_t)K * N * sizeof(float);
    float4 ra[NA], rb[NB];
    // chunk (row0, k0): descriptors whose first byte is A[row0][k0] / B[k0][col0] and whose size is what is left of the matrix (capped at
    // 4 GB - 4: a tile is 128 rows, the cap is never what decides a row of it)
    auto fetch = [&](int row0, int k0) {
        const size_t ao = ((size_t)row0 * K + k0) * sizeof(float), bo = ((size_t)k0 * N + col0) * sizeof(float);
        const size_t ar = ao < a_bytes ? a_bytes - ao : 0, br = bo < b_bytes ? b_bytes - bo : 0;
        const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(A) + ((size_t)row0 * K + k0), 0,
                                                                                (int)(unsigned)(ar < 0xfffffffcULL ? ar : 0xfffffffcULL), 0x00020000);
        const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(B) + ((size_t)k0 * N + col0), 0,
                                                                                (int)(unsigned)(br < 0xfffffffcULL ? br : 0xfffffffcULL), 0x00020000);
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            if constexpr (AV == 4) {
                const u4 v = __builtin_amdgcn_raw_buffer_load_b128(arsrc, voa[j], 0, 0);
                ra[j] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
            } else {
                const u2 lo = __builtin_amdgcn_raw_buffer_load_b64(arsrc, voa[j], 0, 0), hi = __builtin_amdgcn_raw_buffer_load_b64(arsrc, voa[j] + 8, 0, 0);
                ra[j] = make_float4(__uint_as_float(lo[0]), __uint_as_float(lo[1]), __uint_as_float(hi[0]), __uint_as_float(hi[1]));
            }
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const u4 v = __builtin_amdgcn_raw_buffer_load_b128(brsrc, vob[j], 0, 0);
            rb[j] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
        }
    };
    auto stash = [&](float *As, float *Bs, int k0) {
        if (k0 + kBigKC > K) {   // the ragged last chunk (workgroup-uniform): A's k beyond K belongs to the next row, not to nothing
#pragma unroll
            for (int j = 0; j < NA; ++j) {
                const int c = k0 + (((int)threadIdx.x + 256 * j) % KQ) * 4;
                const unsigned m0 = c < K ? 0xffffffffu : 0u, m1 = c + 1 < K ? 0xffffffffu : 0u, m2 = c + 2 < K ? 0xffffffffu : 0u, m3 = c + 3 < K ? 0xffffffffu : 0u;
                ra[j].x = __uint_as_float(__float_as_uint(ra[j].x) & m0); ra[j].y = __uint_as_float(__float_as_uint(ra[j].y) & m1);
                ra[j].z = __uint_as_float(__float_as_uint(ra[j].z) & m2); ra[j].w = __uint_as_float(__float_as_uint(ra[j].w) & m3);
            }
        }
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            float *da = As + la[j];
            da[0] = ra[j].x; da[1] = ra[j].y; da[2] = ra[j].z; da[3] = ra[j].w;
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) *reinterpret_cast<float4 *>(Bs + lb[j]) = rb[j];
    };
    // RBK (32-row blocks of this tile: 4 except at the end of a strip) is a compile-time constant of the burst: with a run-time count every
    // MFMA sat in a basic block of its own behind a scalar branch (ISA of round 4's kernel: 64 one-MFMA blocks, ~190 branch instructions and
    // a wait per chunk), so nothing could be scheduled across them
    auto mma = [&](auto rbk_c, const float *As, const float *Bs) {
        constexpr int RBK = decltype(rbk_c)::value;
        const float *ap = As + (lane & 31) * kBigPA + (lane >> 5);
        const float *bp = Bs + (lane >> 5) * kBigT + 32 * wave + (lane & 31);
        float a_cur[RBK], a_nxt[RBK], b_cur, b_nxt;
#pragma unroll
        for (int i = 0; i < RBK; ++i) a_cur[i] = ap[i * 32 * kBigPA];
        b_cur = bp[0];
#pragma unroll
        for (int t = 0; t < kBigKC / 2; ++t) {
            if (t + 1 < kBigKC / 2) {
#pragma unroll
                for (int i = 0; i < RBK; ++i) a_nxt[i] = ap[i * 32 * kBigPA + 2 * (t + 1)];
                b_nxt = bp[2 * (t + 1) * kBigT];
            }
#pragma unroll
            for (int i = 0; i < RBK; ++i) {
#if GNNAGG_GEMM_SWAP
                // the weight strip rides as the A operand and the 32 rows of X as the B operand: the accumulator is the TRANSPOSED 32 x 32
                // block, so a lane ends up with four CONSECUTIVE COLUMNS of one output row per register quad -- 16-byte stores.  Same
                // products, same k order (a * b = b * a): the same bits.
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(b_cur, a_cur[i], acc[i], 0, 0, 0);
#else
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[i], b_cur, acc[i], 0, 0, 0);
#endif
            }
#pragma unroll
            for (int i = 0; i < RBK; ++i) a_cur[i] = a_nxt[i];
            b_cur = b_nxt;
        }
    };
    const int nchunks = (K + kBigKC - 1) / kBigKC;
    const int ntiles = (nblk + 3) >> 2, total = ntiles * nchunks;
#if !GNNAGG_GEMM_SWAP
    const int colc = (32 * wave + (lane & 31)) * (int)sizeof(float);
#endif
    fetch(blk0 * 32, 0);
    stash(As0, Bs0, 0);
    __syncthreads();
    int c = 0, row0 = blk0 * 32, left = nblk;   // chunk inside the tile, the tile's first row, 32-row blocks from this tile on
    for (int g = 0; g < total; ++g) {
        const float *As = (g & 1) ? As1 : As0, *Bs = (g & 1) ? Bs1 : Bs0;
        const int rbk = left < 4 ? left : 4;
        const bool last_c = c + 1 == nchunks;
        // the next chunk (the last step re-requests its own: unconditional, see k_dense_nn_strip)
        const int nrow0 = last_c && g + 1 < total ? row0 + kBigT : row0;
        const int nk0 = g + 1 < total ? (last_c ? 0 : (c + 1) * kBigKC) : c * kBigKC;
        TL_STAMP(0);
        fetch(nrow0, nk0);
        TL_STAMP(1);
        // (measured on this form and not kept, profiles/r04/gemm.txt: s_setprio low inside the burst / high outside -- the stash and barrier
        // phases shrink, the fetch phase grows, 191.8-205 against 194 us; a second register set with chunk g + 2 in flight -- 195.5; K chunks
        // of 16 with 3 / 4 workgroups per CU -- 193.6 / 207.5; the next chunk's stash folded into the second half of the burst -- the stash
        // phase goes from 19 % to 4 % of the period and the burst grows by as much: 200 against 194 us at the same ratio to rocBLAS)
        if (rbk == 4) mma(std::integral_constant<int, 4>{}, As, Bs);
        else if (rbk == 3) mma(std::integral_constant<int, 3>{}, As, Bs);
        else if (rbk == 2) mma(std::integral_constant<int, 2>{}, As, Bs);
        else mma(std::integral_constant<int, 1>{}, As, Bs);
        TL_STAMP(2);
        if (last_c) {
            // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); a descriptor rebased to
            // C[row0][col0]: one lane offset + a scalar offset per store, rows beyond M fall off its end
            const size_t co = ((size_t)row0 * N + col0) * sizeof(float), c_bytes = (size_t)M * N * sizeof(float);
            const size_t cr = co < c_bytes ? c_bytes - co : 0;
            const __amdgpu_buffer_rsrc_t crsrc = __builtin_amdgcn_make_buffer_rsrc(C + ((size_t)row0 * N + col0), 0,
                                                                                    (int)(unsigned)(cr < 0xfffffffcULL ? cr : 0xfffffffcULL), 0x00020000);
#if GNNAGG_GEMM_SWAP
            // transposed accumulator: row = 32 i + (lane & 31), columns 32 wave + 8 q + 4 (lane >> 5) + 0 .. 3 in registers 4 q .. 4 q + 3
            const int voff = ((lane & 31) * N + 32 * wave + 4 * (lane >> 5)) * (int)sizeof(float);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (i < rbk) {
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) {
                        u4 v;
                        v[0] = __float_as_uint(acc[i][4 * qd]); v[1] = __float_as_uint(acc[i][4 * qd + 1]);
                        v[2] = __float_as_uint(acc[i][4 * qd + 2]); v[3] = __float_as_uint(acc[i][4 * qd + 3]);
                        __builtin_amdgcn_raw_buffer_store_b128(v, crsrc, voff, (32 * i * N + 8 * qd) * (int)sizeof(float), 0);
                    }
                }
            }
            // A store of more than 8 bytes reads its upper data registers a cycle after it issues; the VALU write that follows needs wait
            // states in between.  The compiler inserts them inside a basic block, but here the last 16-byte store falls through a block
            // boundary straight into the zeroing (ISA: buffer_store_dwordx4 v[12:15] / .LBB: v_mov_b32 v15, 0) and a few hundred elements
            // of a 131 072-row product came out as 0.  The wait states are written out.
            asm volatile("s_nop 4" ::: "memory");
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) acc[i][reg] = 0.0f;
            }
#else
            const int voff = 4 * (lane >> 5) * N * (int)sizeof(float) + colc;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (i < rbk) {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg)
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[i][reg]), crsrc, voff,
                                                              (32 * i + (reg & 3) + 8 * (reg >> 2)) * N * (int)sizeof(float), 0);
                }
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) acc[i][reg] = 0.0f;
            }
#endif
        }
        TL_STAMP(3);
        if (g + 1 < total) stash((g & 1) ? As0 : As1, (g & 1) ? Bs0 : Bs1, nk0);
        TL_STAMP(4);
        __syncthreads();
        TL_STAMP(5);
        if (last_c) { c = 0; row0 += kBigT; left -= 4; } else ++c;
    }
}

// k_dense_nn_lean with every chunk requested TWO periods before it is written to LDS, the wait for it counted by hand (round 5).
// Timing variants (profiles/r05/gemm_forms.txt) put the loss of the lean kernel against its compute-only rate on the HBM stream of A
// arriving late: with A served from L2 the same kernel runs at the compute-only rate.  A second register set alone does not buy a
// second period: hipcc's s_waitcnt bookkeeping does not survive the loop back-edge and the barrier -- it waits vmcnt(0) in front of the
// LDS writes, for the chunk requested a moment ago as well -- and a load whose destination is a compiler-visible value may be copied
// (v_mov) before it has landed.  So the producer side is inline assembly on FIXED registers the compiler never sees:
// `amdgpu_num_vgpr(192)` keeps its allocator below v192 (it needs 155), the two register sets are v192-v223 and v224-v255 (named as
// clobbers, so the kernel descriptor still says 256), and the loads, the one wait -- `s_waitcnt vmcnt(8)`: everything but the eight
// newest -- and the LDS writes are written out.  vmcnt also counts the C stores of a tile's last chunk; loads return in order among
// loads, so "at most 8 outstanding" always covers the 8 oldest loads.  Every iteration requests exactly 8 loads (past the end of the
// strip: the last chunk again), so the count is the same everywhere.  Chunk h travels in register set h & 1 and lands in LDS image
// h & 1.  Same arithmetic as k_dense_nn_lean: bit-exact.  16-byte aligned A rows and K % 32 == 0; other shapes stay on k_dense_nn_lean.
// Measured (profiles/r05/gemm_forms.txt): 169 343 x 512 @ 512 x 128 204 -> 192 us (115.6 TF); THREE periods (three sets from v160, the
// compiler squeezed into 160 registers, the image of a chunk a run-time value) 198.6 us -- two is where it pays.
#define GNNAGG_AHEAD_LOAD(REGS, C0, C1, C2, C3, OFF, RSRC) \
    asm volatile("buffer_load_dwordx4 " REGS ", %0, %1, 0 offen" :: "v"(OFF), "s"(RSRC) : "memory", C0, C1, C2, C3)
#define GNNAGG_AHEAD_STASH_A(R0, R1, R2, R3, ADDR) \
    asm volatile("ds_write2_b32 %0, " R0 ", " R1 " offset1:1\n\tds_write2_b32 %0, " R2 ", " R3 " offset0:2 offset1:3" :: "v"(ADDR) : "memory")
#define GNNAGG_AHEAD_STASH_B(REGS, ADDR) asm volatile("ds_write_b128 %0, " REGS :: "v"(ADDR) : "memory")
__global__ __launch_bounds__(256, kBigWgs) __attribute__((amdgpu_num_vgpr(192))) void k_dense_nn_ahead(
    const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ C, int M, int N, int K, int nb32, int nstrips)
{
    extern __shared__ float lds[];
    constexpr int KQ = kBigKC / 4, NA = kBigT * KQ / 256, NB = kBigKC * (kBigT / 4) / 256;
    static_assert(NA == 4 && NB == 4, "the register sets and the hand-counted wait below assume 4 + 4 loads of 16 bytes per chunk");
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) float *lds_f;
    const int col0 = blockIdx.y * kBigT;
    const int q = nb32 / nstrips, extra = nb32 - q * nstrips, sidx = blockIdx.x;
    const int blk0 = sidx * q + (sidx < extra ? sidx : extra), nblk = q + (sidx < extra ? 1 : 0);
    if (nblk == 0) return;
    float *As0 = lds, *As1 = lds + kBigT * kBigPA, *Bs0 = lds + 2 * kBigT * kBigPA, *Bs1 = Bs0 + kBigKC * kBigT;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    // per-thread constants: byte offsets of this thread's pieces inside a chunk (global) and LDS byte addresses inside image 0
    int voa[NA], vob[NB];
    unsigned la0[NA], lb0[NB];
    const unsigned img_a = kBigT * kBigPA * (unsigned)sizeof(float), img_b = kBigKC * kBigT * (unsigned)sizeof(float);   // image 1 = image 0 + this
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int p = (int)threadIdx.x + 256 * j;
        voa[j] = ((p / KQ) * K + (p % KQ) * 4) * (int)sizeof(float);
        la0[j] = (unsigned)(unsigned long long)(lds_f)(As0 + (p / KQ) * kBigPA + (p % KQ) * 4);
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        vob[j] = (((int)(threadIdx.x >> 5) + 8 * j) * N + (int)(threadIdx.x & 31) * 4) * (int)sizeof(float);
        lb0[j] = (unsigned)(unsigned long long)(lds_f)(Bs0 + ((int)(threadIdx.x >> 5) + 8 * j) * kBigT + (int)(threadIdx.x & 31) * 4);
    }
    const size_t a_bytes = (size_t)M * K * sizeof(float), b_bytes = (size_t)K * N * sizeof(float);
    // a raw buffer descriptor by hand (base, stride 0, bytes, the flags __builtin_amdgcn_make_buffer_rsrc is given elsewhere in this file)
    auto rsrc_of = [](const float *base, size_t bytes) -> u4 {
        const unsigned long long a = (unsigned long long)base;
        u4 r;
        r[0] = (unsigned)a; r[1] = (unsigned)(a >> 32) & 0xffffu; r[2] = (unsigned)(bytes < 0xfffffffcULL ? bytes : 0xfffffffcULL); r[3] = 0x00020000u;
        return r;
    };
    const int nchunks = K / kBigKC;
    const int ntiles = (nblk + 3) >> 2, total = ntiles * nchunks;
    // the next chunk to REQUEST: (fc, frow0), index fi; past the last chunk of the strip the last one is requested again
    int fc = 0, frow0 = blk0 * 32, fi = 0;
#define GNNAGG_AHEAD_FETCH(SET)                                                                                                                       \
    {                                                                                                                                                 \
        const int k0_ = fc * kBigKC;                                                                                                                  \
        const size_t ao_ = ((size_t)frow0 * K + k0_) * sizeof(float), bo_ = ((size_t)k0_ * N + col0) * sizeof(float);                                 \
        const u4 ar_ = rsrc_of(A + ((size_t)frow0 * K + k0_), ao_ < a_bytes ? a_bytes - ao_ : 0);                                                     \
        const u4 br_ = rsrc_of(B + ((size_t)k0_ * N + col0), bo_ < b_bytes ? b_bytes - bo_ : 0);                                                      \
        if (SET == 0) {                                                                                                                               \
            GNNAGG_AHEAD_LOAD("v[192:195]", "v192", "v193", "v194", "v195", voa[0], ar_);                                                             \
            GNNAGG_AHEAD_LOAD("v[196:199]", "v196", "v197", "v198", "v199", voa[1], ar_);                                                             \
            GNNAGG_AHEAD_LOAD("v[200:203]", "v200", "v201", "v202", "v203", voa[2], ar_);                                                             \
            GNNAGG_AHEAD_LOAD("v[204:207]", "v204", "v205", "v206", "v207", voa[3], ar_);                                                             \
            GNNAGG_AHEAD_LOAD("v[208:211]", "v208", "v209", "v210", "v211", vob[0], br_);                                                             \
            GNNAGG_AHEAD_LOAD("v[212:215]", "v212", "v213", "v214", "v215", vob[1], br_);                                                             \
            GNNAGG_AHEAD_LOAD("v[216:219]", "v216", "v217", "v218", "v219", vob[2], br_);                                                             \
            GNNAGG_AHEAD_LOAD("v[220:223]", "v220", "v221", "v222", "v223", vob[3], br_);                                                             \
        } else {                                                                                                                                      \
            GNNAGG_AHEAD_LOAD("v[224:227]", "v224", "v225", "v226", "v227", voa[0], ar_);                                                             \
            GNNAGG_AHEAD_LOAD("v[228:231]", "v228", "v229", "v230", "v231", voa[1], ar_);                                                             \
            GNNAGG_AHEAD_LOAD("v[232:235]", "v232", "v233", "v234", "v235", voa[2], ar_);                                                             \
            GNNAGG_AHEAD_LOAD("v[236:239]", "v236", "v237", "v238", "v239", voa[3], ar_);                                                             \
            GNNAGG_AHEAD_LOAD("v[240:243]", "v240", "v241", "v242", "v243", vob[0], br_);                                                             \
            GNNAGG_AHEAD_LOAD("v[244:247]", "v244", "v245", "v246", "v247", vob[1], br_);                                                             \
            GNNAGG_AHEAD_LOAD("v[248:251]", "v248", "v249", "v250", "v251", vob[2], br_);                                                             \
            GNNAGG_AHEAD_LOAD("v[252:255]", "v252", "v253", "v254", "v255", vob[3], br_);                                                             \
        }                                                                                                                                             \
        if (fi + 1 < total) { ++fi; if (fc + 1 == nchunks) { fc = 0; frow0 += kBigT; } else ++fc; }                                                   \
    }
    // set SET (landed: the caller waited) -> LDS image SET; then the LDS writes are drained so that the set can be requested into again
#define GNNAGG_AHEAD_STASH(SET)                                                                                                                       \
    {                                                                                                                                                 \
        if (SET == 0) {                                                                                                                               \
            GNNAGG_AHEAD_STASH_A("v192", "v193", "v194", "v195", la0[0]);                                                                             \
            GNNAGG_AHEAD_STASH_A("v196", "v197", "v198", "v199", la0[1]);                                                                             \
            GNNAGG_AHEAD_STASH_A("v200", "v201", "v202", "v203", la0[2]);                                                                             \
            GNNAGG_AHEAD_STASH_A("v204", "v205", "v206", "v207", la0[3]);                                                                             \
            GNNAGG_AHEAD_STASH_B("v[208:211]", lb0[0]);                                                                                               \
            GNNAGG_AHEAD_STASH_B("v[212:215]", lb0[1]);                                                                                               \
            GNNAGG_AHEAD_STASH_B("v[216:219]", lb0[2]);                                                                                               \
            GNNAGG_AHEAD_STASH_B("v[220:223]", lb0[3]);                                                                                               \
        } else {                                                                                                                                      \
            GNNAGG_AHEAD_STASH_A("v224", "v225", "v226", "v227", la0[0] + img_a);                                                                     \
            GNNAGG_AHEAD_STASH_A("v228", "v229", "v230", "v231", la0[1] + img_a);                                                                     \
            GNNAGG_AHEAD_STASH_A("v232", "v233", "v234", "v235", la0[2] + img_a);                                                                     \
            GNNAGG_AHEAD_STASH_A("v236", "v237", "v238", "v239", la0[3] + img_a);                                                                     \
            GNNAGG_AHEAD_STASH_B("v[240:243]", lb0[0] + img_b);                                                                                       \
            GNNAGG_AHEAD_STASH_B("v[244:247]", lb0[1] + img_b);                                                                                       \
            GNNAGG_AHEAD_STASH_B("v[248:251]", lb0[2] + img_b);                                                                                       \
            GNNAGG_AHEAD_STASH_B("v[252:255]", lb0[3] + img_b);                                                                                       \
        }                                                                                                                                             \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                                            \
    }
    auto mma = [&](auto rbk_c, const float *As, const float *Bs) __attribute__((always_inline)) {
        constexpr int RBK = decltype(rbk_c)::value;
        const float *ap = As + (lane & 31) * kBigPA + (lane >> 5);
        const float *bp = Bs + (lane >> 5) * kBigT + 32 * wave + (lane & 31);
#if GNNAGG_GEMM_AHEAD_PIPE
        // two k-steps (a "pair": 2 RBK MFMAs) per stage; the operand reads of pair p + 1 are issued BEFORE the MFMAs of pair p (left to
        // itself the compiler issues them behind the pair's last MFMA and waits for them in front of the next one)
        float a[2][RBK][2], b[2][2];
#define GNNAGG_RD(BUF, PR)                                                                                                    \
        {                                                                                                                     \
            _Pragma("unroll") for (int i = 0; i < RBK; ++i) {                                                                 \
                a[BUF][i][0] = ap[i * 32 * kBigPA + 4 * (PR)]; a[BUF][i][1] = ap[i * 32 * kBigPA + 4 * (PR) + 2];             \
            }                                                                                                                 \
            b[BUF][0] = bp[4 * (PR) * kBigT]; b[BUF][1] = bp[(4 * (PR) + 2) * kBigT];                                         \
        }
        GNNAGG_RD(0, 0)
#pragma unroll
        for (int pr = 0; pr < kBigKC / 4; ++pr) {
            if (pr + 1 < kBigKC / 4) GNNAGG_RD((pr + 1) & 1, pr + 1)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < RBK; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[pr & 1][i][kk], b[pr & 1][kk], acc[i], 0, 0, 0);
#if GNNAGG_GEMM_AHEAD_PIPE == 2   // the next pair's reads spread between this pair's MFMAs, one behind each of the first RBK + 1
            if (pr + 1 < kBigKC / 4) {
#pragma unroll
                for (int r = 0; r < RBK + 1; ++r) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 2 * RBK - (RBK + 1), 0);
            } else {
                __builtin_amdgcn_sched_group_barrier(0x008, 2 * RBK, 0);
            }
#else
            if (pr + 1 < kBigKC / 4) __builtin_amdgcn_sched_group_barrier(0x100, RBK + 1, 0);   // DS reads of the next pair first ...
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * RBK, 0);                              // ... then this pair's MFMAs
#endif
        }
#undef GNNAGG_RD
#else
        float a_cur[RBK], a_nxt[RBK], b_cur, b_nxt;
#pragma unroll
        for (int i = 0; i < RBK; ++i) a_cur[i] = ap[i * 32 * kBigPA];
        b_cur = bp[0];
#pragma unroll
        for (int t = 0; t < kBigKC / 2; ++t) {
            if (t + 1 < kBigKC / 2) {
#pragma unroll
                for (int i = 0; i < RBK; ++i) a_nxt[i] = ap[i * 32 * kBigPA + 2 * (t + 1)];
                b_nxt = bp[2 * (t + 1) * kBigT];
            }
#pragma unroll
            for (int i = 0; i < RBK; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[i], b_cur, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < RBK; ++i) a_cur[i] = a_nxt[i];
            b_cur = b_nxt;
        }
#endif
    };
    const int colc = (32 * wave + (lane & 31)) * (int)sizeof(float);
    int c = 0, row0 = blk0 * 32, left = nblk;   // the chunk being multiplied
    // prologue: chunks 0 and 1 requested, chunk 0 landed and written, chunk 2 requested
    GNNAGG_AHEAD_FETCH(0)
    GNNAGG_AHEAD_FETCH(1)
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    GNNAGG_AHEAD_STASH(0)
    GNNAGG_AHEAD_FETCH(0)
    __syncthreads();
    for (int g0 = 0; g0 < total; g0 += 2) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {   // unrolled: the register set and the LDS image of a chunk (h & 1) are compile-time constants
            const int g = g0 + half;
            if (g >= total) break;
            // ---- multiply chunk g (image g & 1)
            const float *As = half ? As1 : As0, *Bs = half ? Bs1 : Bs0;
            const int rbk = left < 4 ? left : 4;
            const bool last_c = c + 1 == nchunks;
            if (rbk == 4) mma(std::integral_constant<int, 4>{}, As, Bs);
            else if (rbk == 3) mma(std::integral_constant<int, 3>{}, As, Bs);
            else if (rbk == 2) mma(std::integral_constant<int, 2>{}, As, Bs);
            else mma(std::integral_constant<int, 1>{}, As, Bs);
            const int srow0 = row0, srbk = rbk;   // (the tile whose last chunk this is: its C stores go out BEHIND the request below)
            if (last_c) { c = 0; row0 += kBigT; left -= 4; } else ++c;
            // ---- chunk g + 1 (requested two periods ago; everything but the 8 newest loads has landed) goes to the other image, and the
            //      set it frees takes the request of chunk g + 3 (past the end: the last chunk once more -- the count stays the same)
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            if (half == 0) {
                if (g + 1 < total) GNNAGG_AHEAD_STASH(1)
                GNNAGG_AHEAD_FETCH(1)
            } else {
                if (g + 1 < total) GNNAGG_AHEAD_STASH(0)
                GNNAGG_AHEAD_FETCH(0)
            }
            if (last_c) {
                // The tile's C stores, behind the wait and the request: in front of the wait they were its 64 newest operations, and
                // "all but the 8 newest" then meant the stores just issued AND the chunk requested a period ago.  Here the next wait finds
                // them a whole period old.  (Still safe: at most 8 outstanding operations cannot be the 8 needed loads unless the 8
                // younger loads are outstanding too -- loads return in order.)
                // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); descriptor rebased to C[srow0][col0]
                const size_t co = ((size_t)srow0 * N + col0) * sizeof(float), c_bytes = (size_t)M * N * sizeof(float);
                const size_t cr = co < c_bytes ? c_bytes - co : 0;
                const __amdgpu_buffer_rsrc_t crsrc = __builtin_amdgcn_make_buffer_rsrc(C + ((size_t)srow0 * N + col0), 0,
                                                                                        (int)(unsigned)(cr < 0xfffffffcULL ? cr : 0xfffffffcULL), 0x00020000);
                const int voff = 4 * (lane >> 5) * N * (int)sizeof(float) + colc;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (i < srbk) {
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg)
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[i][reg]), crsrc, voff,
                                                                  (32 * i + (reg & 3) + 8 * (reg >> 2)) * N * (int)sizeof(float), 0);
                    }
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) acc[i][reg] = 0.0f;
                }
            }
            __syncthreads();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
#undef GNNAGG_AHEAD_FETCH
#undef GNNAGG_AHEAD_STASH
#undef GNNAGG_AHEAD_LOAD
#undef GNNAGG_AHEAD_STASH_A
#undef GNNAGG_AHEAD_STASH_B

// The same kernel for A rows that are only 8-byte aligned and any even K (the 602-wide layer): a piece is two 8-byte loads, 12 loads per
// chunk (`s_waitcnt vmcnt(12)`), and the last chunk of a ragged K is masked in the registers before it goes to LDS (A's k beyond K belongs
// to the next row, not to nothing; B's rows beyond K are outside its descriptor).
#define GNNAGG_AHEAD_LOAD(REGS, C0, C1, C2, C3, OFF, RSRC) \
    asm volatile("buffer_load_dwordx4 " REGS ", %0, %1, 0 offen" :: "v"(OFF), "s"(RSRC) : "memory", C0, C1, C2, C3)
#define GNNAGG_AHEAD_LOAD2(C0, C1, C2, C3, OFF, RSRC)                                                                              \
    asm volatile("buffer_load_dwordx2 v[" C0 ":" C1 "], %0, %1, 0 offen\n\tbuffer_load_dwordx2 v[" C2 ":" C3 "], %0, %1, 0 offen offset:8" \
                 :: "v"(OFF), "s"(RSRC) : "memory", "v" C0, "v" C1, "v" C2, "v" C3)
#define GNNAGG_AHEAD_MASK(C0, C1, C2, C3, CBASE)                                                                                    \
    {                                                                                                                              \
        const int c_ = (CBASE);                                                                                                    \
        const unsigned m0_ = c_ < K ? 0xffffffffu : 0u, m1_ = c_ + 1 < K ? 0xffffffffu : 0u, m2_ = c_ + 2 < K ? 0xffffffffu : 0u,   \
                       m3_ = c_ + 3 < K ? 0xffffffffu : 0u;                                                                         \
        asm volatile("v_and_b32 v" C0 ", %0, v" C0 "\n\tv_and_b32 v" C1 ", %1, v" C1 "\n\tv_and_b32 v" C2 ", %2, v" C2 "\n\tv_and_b32 v" C3 ", %3, v" C3 \
                     :: "v"(m0_), "v"(m1_), "v"(m2_), "v"(m3_) : "memory", "v" C0, "v" C1, "v" C2, "v" C3);                          \
    }
#define GNNAGG_AHEAD_STASH_A(R0, R1, R2, R3, ADDR) \
    asm volatile("ds_write2_b32 %0, " R0 ", " R1 " offset1:1\n\tds_write2_b32 %0, " R2 ", " R3 " offset0:2 offset1:3" :: "v"(ADDR) : "memory")
#define GNNAGG_AHEAD_STASH_B(REGS, ADDR) asm volatile("ds_write_b128 %0, " REGS :: "v"(ADDR) : "memory")
__global__ __launch_bounds__(256, kBigWgs) __attribute__((amdgpu_num_vgpr(192))) void k_dense_nn_ahead2(
    const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ C, int M, int N, int K, int nb32, int nstrips)
{
    extern __shared__ float lds[];
    constexpr int KQ = kBigKC / 4, NA = kBigT * KQ / 256, NB = kBigKC * (kBigT / 4) / 256;
    static_assert(NA == 4 && NB == 4, "the register sets and the hand-counted wait below assume 4 pieces of A (two 8-byte loads each) + 4 of B per chunk");
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) float *lds_f;
    const int col0 = blockIdx.y * kBigT;
    const int q = nb32 / nstrips, extra = nb32 - q * nstrips, sidx = blockIdx.x;
    const int blk0 = sidx * q + (sidx < extra ? sidx : extra), nblk = q + (sidx < extra ? 1 : 0);
    if (nblk == 0) return;
    float *As0 = lds, *As1 = lds + kBigT * kBigPA, *Bs0 = lds + 2 * kBigT * kBigPA, *Bs1 = Bs0 + kBigKC * kBigT;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    // per-thread constants: byte offsets of this thread's pieces inside a chunk (global) and LDS byte addresses inside image 0
    int voa[NA], vob[NB];
    unsigned la0[NA], lb0[NB];
    const unsigned img_a = kBigT * kBigPA * (unsigned)sizeof(float), img_b = kBigKC * kBigT * (unsigned)sizeof(float);   // image 1 = image 0 + this
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int p = (int)threadIdx.x + 256 * j;
        voa[j] = ((p / KQ) * K + (p % KQ) * 4) * (int)sizeof(float);
        la0[j] = (unsigned)(unsigned long long)(lds_f)(As0 + (p / KQ) * kBigPA + (p % KQ) * 4);
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        vob[j] = (((int)(threadIdx.x >> 5) + 8 * j) * N + (int)(threadIdx.x & 31) * 4) * (int)sizeof(float);
        lb0[j] = (unsigned)(unsigned long long)(lds_f)(Bs0 + ((int)(threadIdx.x >> 5) + 8 * j) * kBigT + (int)(threadIdx.x & 31) * 4);
    }
    const size_t a_bytes = (size_t)M * K * sizeof(float), b_bytes = (size_t)K * N * sizeof(float);
    // a raw buffer descriptor by hand (base, stride 0, bytes, the flags __builtin_amdgcn_make_buffer_rsrc is given elsewhere in this file)
    auto rsrc_of = [](const float *base, size_t bytes) -> u4 {
        const unsigned long long a = (unsigned long long)base;
        u4 r;
        r[0] = (unsigned)a; r[1] = (unsigned)(a >> 32) & 0xffffu; r[2] = (unsigned)(bytes < 0xfffffffcULL ? bytes : 0xfffffffcULL); r[3] = 0x00020000u;
        return r;
    };
    const int nchunks = (K + kBigKC - 1) / kBigKC;
    const int ntiles = (nblk + 3) >> 2, total = ntiles * nchunks;
    // the next chunk to REQUEST: (fc, frow0), index fi; past the last chunk of the strip the last one is requested again
    int fc = 0, frow0 = blk0 * 32, fi = 0;
    int sc = 0;   // chunk-in-tile of the next chunk to WRITE to LDS (its k0 decides the ragged mask)
#define GNNAGG_AHEAD_FETCH(SET)                                                                                                                       \
    {                                                                                                                                                 \
        const int k0_ = fc * kBigKC;                                                                                                                  \
        const size_t ao_ = ((size_t)frow0 * K + k0_) * sizeof(float), bo_ = ((size_t)k0_ * N + col0) * sizeof(float);                                 \
        const u4 ar_ = rsrc_of(A + ((size_t)frow0 * K + k0_), ao_ < a_bytes ? a_bytes - ao_ : 0);                                                     \
        const u4 br_ = rsrc_of(B + ((size_t)k0_ * N + col0), bo_ < b_bytes ? b_bytes - bo_ : 0);                                                      \
        if (SET == 0) {                                                                                                                               \
            GNNAGG_AHEAD_LOAD2("192", "193", "194", "195", voa[0], ar_);                                                             \
            GNNAGG_AHEAD_LOAD2("196", "197", "198", "199", voa[1], ar_);                                                             \
            GNNAGG_AHEAD_LOAD2("200", "201", "202", "203", voa[2], ar_);                                                             \
            GNNAGG_AHEAD_LOAD2("204", "205", "206", "207", voa[3], ar_);                                                             \
            GNNAGG_AHEAD_LOAD("v[208:211]", "v208", "v209", "v210", "v211", vob[0], br_);                                                             \
            GNNAGG_AHEAD_LOAD("v[212:215]", "v212", "v213", "v214", "v215", vob[1], br_);                                                             \
            GNNAGG_AHEAD_LOAD("v[216:219]", "v216", "v217", "v218", "v219", vob[2], br_);                                                             \
            GNNAGG_AHEAD_LOAD("v[220:223]", "v220", "v221", "v222", "v223", vob[3], br_);                                                             \
        } else {                                                                                                                                      \
            GNNAGG_AHEAD_LOAD2("224", "225", "226", "227", voa[0], ar_);                                                             \
            GNNAGG_AHEAD_LOAD2("228", "229", "230", "231", voa[1], ar_);                                                             \
            GNNAGG_AHEAD_LOAD2("232", "233", "234", "235", voa[2], ar_);                                                             \
            GNNAGG_AHEAD_LOAD2("236", "237", "238", "239", voa[3], ar_);                                                             \
            GNNAGG_AHEAD_LOAD("v[240:243]", "v240", "v241", "v242", "v243", vob[0], br_);                                                             \
            GNNAGG_AHEAD_LOAD("v[244:247]", "v244", "v245", "v246", "v247", vob[1], br_);                                                             \
            GNNAGG_AHEAD_LOAD("v[248:251]", "v248", "v249", "v250", "v251", vob[2], br_);                                                             \
            GNNAGG_AHEAD_LOAD("v[252:255]", "v252", "v253", "v254", "v255", vob[3], br_);                                                             \
        }                                                                                                                                             \
        if (fi + 1 < total) { ++fi; if (fc + 1 == nchunks) { fc = 0; frow0 += kBigT; } else ++fc; }                                                   \
    }
    // set SET (landed: the caller waited) -> LDS image SET; then the LDS writes are drained so that the set can be requested into again
#define GNNAGG_AHEAD_STASH(SET, K0)                                                                                                                   \
    {                                                                                                                                                 \
        if ((K0) + kBigKC > K) {   /* the ragged last chunk (workgroup-uniform) */                                                                    \
            if (SET == 0) {                                                                                                                           \
            GNNAGG_AHEAD_MASK("192", "193", "194", "195", (K0) + (((int)threadIdx.x + 256 * 0) % KQ) * 4)                                             \
            GNNAGG_AHEAD_MASK("196", "197", "198", "199", (K0) + (((int)threadIdx.x + 256 * 1) % KQ) * 4)                                             \
            GNNAGG_AHEAD_MASK("200", "201", "202", "203", (K0) + (((int)threadIdx.x + 256 * 2) % KQ) * 4)                                             \
            GNNAGG_AHEAD_MASK("204", "205", "206", "207", (K0) + (((int)threadIdx.x + 256 * 3) % KQ) * 4)                                             \
            } else {                                                                                                                                  \
            GNNAGG_AHEAD_MASK("224", "225", "226", "227", (K0) + (((int)threadIdx.x + 256 * 0) % KQ) * 4)                                             \
            GNNAGG_AHEAD_MASK("228", "229", "230", "231", (K0) + (((int)threadIdx.x + 256 * 1) % KQ) * 4)                                             \
            GNNAGG_AHEAD_MASK("232", "233", "234", "235", (K0) + (((int)threadIdx.x + 256 * 2) % KQ) * 4)                                             \
            GNNAGG_AHEAD_MASK("236", "237", "238", "239", (K0) + (((int)threadIdx.x + 256 * 3) % KQ) * 4)                                             \
            }                                                                                                                                         \
        }                                                                                                                                             \
        if (SET == 0) {                                                                                                                               \
            GNNAGG_AHEAD_STASH_A("v192", "v193", "v194", "v195", la0[0]);                                                                             \
            GNNAGG_AHEAD_STASH_A("v196", "v197", "v198", "v199", la0[1]);                                                                             \
            GNNAGG_AHEAD_STASH_A("v200", "v201", "v202", "v203", la0[2]);                                                                             \
            GNNAGG_AHEAD_STASH_A("v204", "v205", "v206", "v207", la0[3]);                                                                             \
            GNNAGG_AHEAD_STASH_B("v[208:211]", lb0[0]);                                                                                               \
            GNNAGG_AHEAD_STASH_B("v[212:215]", lb0[1]);                                                                                               \
            GNNAGG_AHEAD_STASH_B("v[216:219]", lb0[2]);                                                                                               \
            GNNAGG_AHEAD_STASH_B("v[220:223]", lb0[3]);                                                                                               \
        } else {                                                                                                                                      \
            GNNAGG_AHEAD_STASH_A("v224", "v225", "v226", "v227", la0[0] + img_a);                                                                     \
            GNNAGG_AHEAD_STASH_A("v228", "v229", "v230", "v231", la0[1] + img_a);                                                                     \
            GNNAGG_AHEAD_STASH_A("v232", "v233", "v234", "v235", la0[2] + img_a);                                                                     \
            GNNAGG_AHEAD_STASH_A("v236", "v237", "v238", "v239", la0[3] + img_a);                                                                     \
            GNNAGG_AHEAD_STASH_B("v[240:243]", lb0[0] + img_b);                                                                                       \
            GNNAGG_AHEAD_STASH_B("v[244:247]", lb0[1] + img_b);                                                                                       \
            GNNAGG_AHEAD_STASH_B("v[248:251]", lb0[2] + img_b);                                                                                       \
            GNNAGG_AHEAD_STASH_B("v[252:255]", lb0[3] + img_b);                                                                                       \
        }                                                                                                                                             \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                                            \
    }
    auto mma = [&](auto rbk_c, const float *As, const float *Bs) __attribute__((always_inline)) {
        constexpr int RBK = decltype(rbk_c)::value;
        const float *ap = As + (lane & 31) * kBigPA + (lane >> 5);
        const float *bp = Bs + (lane >> 5) * kBigT + 32 * wave + (lane & 31);
#if GNNAGG_GEMM_AHEAD_PIPE
        // two k-steps (a "pair": 2 RBK MFMAs) per stage; the operand reads of pair p + 1 are issued BEFORE the MFMAs of pair p (left to
        // itself the compiler issues them behind the pair's last MFMA and waits for them in front of the next one)
        float a[2][RBK][2], b[2][2];
#define GNNAGG_RD(BUF, PR)                                                                                                    \
        {                                                                                                                     \
            _Pragma("unroll") for (int i = 0; i < RBK; ++i) {                                                                 \
                a[BUF][i][0] = ap[i * 32 * kBigPA + 4 * (PR)]; a[BUF][i][1] = ap[i * 32 * kBigPA + 4 * (PR) + 2];             \
            }                                                                                                                 \
            b[BUF][0] = bp[4 * (PR) * kBigT]; b[BUF][1] = bp[(4 * (PR) + 2) * kBigT];                                         \
        }
        GNNAGG_RD(0, 0)
#pragma unroll
        for (int pr = 0; pr < kBigKC / 4; ++pr) {
            if (pr + 1 < kBigKC / 4) GNNAGG_RD((pr + 1) & 1, pr + 1)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < RBK; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[pr & 1][i][kk], b[pr & 1][kk], acc[i], 0, 0, 0);
#if GNNAGG_GEMM_AHEAD_PIPE == 2   // the next pair's reads spread between this pair's MFMAs, one behind each of the first RBK + 1
            if (pr + 1 < kBigKC / 4) {
#pragma unroll
                for (int r = 0; r < RBK + 1; ++r) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 2 * RBK - (RBK + 1), 0);
            } else {
                __builtin_amdgcn_sched_group_barrier(0x008, 2 * RBK, 0);
            }
#else
            if (pr + 1 < kBigKC / 4) __builtin_amdgcn_sched_group_barrier(0x100, RBK + 1, 0);   // DS reads of the next pair first ...
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * RBK, 0);                              // ... then this pair's MFMAs
#endif
        }
#undef GNNAGG_RD
#else
        float a_cur[RBK], a_nxt[RBK], b_cur, b_nxt;
#pragma unroll
        for (int i = 0; i < RBK; ++i) a_cur[i] = ap[i * 32 * kBigPA];
        b_cur = bp[0];
#pragma unroll
        for (int t = 0; t < kBigKC / 2; ++t) {
            if (t + 1 < kBigKC / 2) {
#pragma unroll
                for (int i = 0; i < RBK; ++i) a_nxt[i] = ap[i * 32 * kBigPA + 2 * (t + 1)];
                b_nxt = bp[2 * (t + 1) * kBigT];
            }
#pragma unroll
            for (int i = 0; i < RBK; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[i], b_cur, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < RBK; ++i) a_cur[i] = a_nxt[i];
            b_cur = b_nxt;
        }
#endif
    };
    const int colc = (32 * wave + (lane & 31)) * (int)sizeof(float);
    int c = 0, row0 = blk0 * 32, left = nblk;   // the chunk being multiplied
    // prologue: chunks 0 and 1 requested, chunk 0 landed and written, chunk 2 requested
    GNNAGG_AHEAD_FETCH(0)
    GNNAGG_AHEAD_FETCH(1)
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    GNNAGG_AHEAD_STASH(0, 0)
    sc = nchunks > 1 ? 1 : 0;
    GNNAGG_AHEAD_FETCH(0)
    __syncthreads();
    for (int g0 = 0; g0 < total; g0 += 2) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {   // unrolled: the register set and the LDS image of a chunk (h & 1) are compile-time constants
            const int g = g0 + half;
            if (g >= total) break;
            // ---- multiply chunk g (image g & 1)
            const float *As = half ? As1 : As0, *Bs = half ? Bs1 : Bs0;
            const int rbk = left < 4 ? left : 4;
            const bool last_c = c + 1 == nchunks;
            if (rbk == 4) mma(std::integral_constant<int, 4>{}, As, Bs);
            else if (rbk == 3) mma(std::integral_constant<int, 3>{}, As, Bs);
            else if (rbk == 2) mma(std::integral_constant<int, 2>{}, As, Bs);
            else mma(std::integral_constant<int, 1>{}, As, Bs);
            const int srow0 = row0, srbk = rbk;   // (the tile whose last chunk this is: its C stores go out BEHIND the request below)
            if (last_c) { c = 0; row0 += kBigT; left -= 4; } else ++c;
            // ---- chunk g + 1 (requested two periods ago; everything but the 8 newest loads has landed) goes to the other image, and the
            //      set it frees takes the request of chunk g + 3 (past the end: the last chunk once more -- the count stays the same)
            asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            if (half == 0) {
                if (g + 1 < total) { GNNAGG_AHEAD_STASH(1, sc * kBigKC) sc = sc + 1 == nchunks ? 0 : sc + 1; }
                GNNAGG_AHEAD_FETCH(1)
            } else {
                if (g + 1 < total) { GNNAGG_AHEAD_STASH(0, sc * kBigKC) sc = sc + 1 == nchunks ? 0 : sc + 1; }
                GNNAGG_AHEAD_FETCH(0)
            }
            if (last_c) {
                // The tile's C stores, behind the wait and the request: in front of the wait they were its 64 newest operations, and
                // "all but the 8 newest" then meant the stores just issued AND the chunk requested a period ago.  Here the next wait finds
                // them a whole period old.  (Still safe: at most 8 outstanding operations cannot be the 8 needed loads unless the 8
                // younger loads are outstanding too -- loads return in order.)
                // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); descriptor rebased to C[srow0][col0]
                const size_t co = ((size_t)srow0 * N + col0) * sizeof(float), c_bytes = (size_t)M * N * sizeof(float);
                const size_t cr = co < c_bytes ? c_bytes - co : 0;
                const __amdgpu_buffer_rsrc_t crsrc = __builtin_amdgcn_make_buffer_rsrc(C + ((size_t)srow0 * N + col0), 0,
                                                                                        (int)(unsigned)(cr < 0xfffffffcULL ? cr : 0xfffffffcULL), 0x00020000);
                const int voff = 4 * (lane >> 5) * N * (int)sizeof(float) + colc;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (i < srbk) {
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg)
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[i][reg]), crsrc, voff,
                                                                  (32 * i + (reg & 3) + 8 * (reg >> 2)) * N * (int)sizeof(float), 0);
                    }
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) acc[i][reg] = 0.0f;
                }
            }
            __syncthreads();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
#undef GNNAGG_AHEAD_FETCH
#undef GNNAGG_AHEAD_STASH
#undef GNNAGG_AHEAD_LOAD
#undef GNNAGG_AHEAD_STASH_A
#undef GNNAGG_AHEAD_STASH_B
#undef GNNAGG_AHEAD_LOAD2
#undef GNNAGG_AHEAD_MASK

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
        // measured against k_dense_nn (N = 32): K = 128: M = 300 k 48.2 vs 46.4 us, 600 k 92.4 vs 99.9, 1.2 M 169 vs 184,
        // 2.45 M 307 vs 352; K = 100, M = 2.45 M: 268 vs 363 (torch.mm 427); K = 64 loses at every M -> large M, wide K only
        if (K > 64 && K <= 128 && (K & 3) == 0 && M >= 500000 && ((uintptr_t)A & 15) == 0) {
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
        return launch_zero_words(C, (size_t)M * N, stream);
    }
    {
        if (N > 64 && M >= 1024 && (size_t)kBigT * N * sizeof(float) < 0x7fffffffULL) {   // wide outputs: persistent strips of 128 x 128 tiles, A read once
            const size_t lds = (size_t)(2 * kBigT * kBigPA + 2 * kBigKC * kBigT) * sizeof(float);
            const bool bvec = (N & 3) == 0 && N >= 4 && ((uintptr_t)B & 15) == 0;
            const int av = !bvec ? 1 : ((K & 3) == 0 && K >= 4 && ((uintptr_t)A & 15) == 0) ? 4 : ((K & 1) == 0 && K >= 2 && ((uintptr_t)A & 7) == 0) ? 2 : 1;
            // the grid is what the chip holds at a time: kBigWgs workgroups per CU, shared by the column tiles
            const int ncol = ceil_div(N, kBigT), slots = std::max(1, kBigWgs * device_cu_count() / ncol);
            const int nb32 = ceil_div(M, 32), nstrips = std::min(slots, ceil_div(nb32, 2));
            const dim3 sgrid(nstrips, ncol);
#define WIDE_CALL(KERNEL_)                                                                                                              \
            {                                                                                                                           \
                static OncePerDevice attr_ok;                                                                                           \
                if (attr_ok.first()) {                                                                                                  \
                    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&KERNEL_), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
                    attr_ok.done();                                                                                                     \
                }                                                                                                                       \
                hipLaunchKernelGGL(KERNEL_, sgrid, dim3(256), lds, stream, A, B, C, M, N, K, nb32, nstrips);                            \
            }
            // lean form: whole 128-column tiles, 16-byte aligned B rows, A rows 16- or 8-byte aligned, row pitches inside 32-bit offsets
            const bool lean = (N % kBigT) == 0 && bvec && av >= 2 && (size_t)kBigT * K * sizeof(float) < 0x7fffffffULL &&
                              (size_t)kBigKC * N * sizeof(float) < 0x7fffffffULL;
#if GNNAGG_GEMM_AHEAD
            if (lean && av == 4 && (K % kBigKC) == 0) WIDE_CALL(k_dense_nn_ahead)
            else if (lean && av == 2) WIDE_CALL(k_dense_nn_ahead2)
            else
#endif
            if (lean && av == 4) WIDE_CALL(k_dense_nn_lean<4>)
            else if (lean) WIDE_CALL(k_dense_nn_lean<2>)
            else if (av == 4) WIDE_CALL(k_dense_nn_strip<4>)
            else if (av == 2) WIDE_CALL(k_dense_nn_strip<2>)
            else WIDE_CALL(k_dense_nn_strip<1>)
#undef WIDE_CALL
            HIP_TRY(hipGetLastError());
            return GNNAGG_OK;
        }
    }
    const dim3 grid(ceil_div(M, kGemmRows), ceil_div(N, kGemmCols));
    if (K % kGemmKC == 0 && K <= 4 * kGemmKC && ((uintptr_t)A & 15) == 0 && (size_t)kGemmRows * K * sizeof(float) < 0x7fffffffULL) {
        // every chunk of a tile requested up front (k_dense_nn_up); 33 .. 64 columns: both 32-column blocks in one workgroup
        const bool two = N > kGemmCols && N <= 2 * kGemmCols;
        const dim3 g(ceil_div(M, kGemmRows), two ? 1 : ceil_div(N, kGemmCols));
#define UP_CALL(NCH_) \
        { if (two) hipLaunchKernelGGL((k_dense_nn_up<NCH_, 2>), g, dim3(256), 0, stream, A, B, C, M, N); \
          else hipLaunchKernelGGL((k_dense_nn_up<NCH_, 1>), g, dim3(256), 0, stream, A, B, C, M, N); }
        switch (K / kGemmKC) {
            case 1: UP_CALL(1) break;
            case 2: UP_CALL(2) break;
            case 3: UP_CALL(3) break;
            default: UP_CALL(4) break;
        }
#undef UP_CALL
        HIP_TRY(hipGetLastError());
        return GNNAGG_OK;
    }
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

// GAT halo rows: one exchange carries the feature row AND the attention terms of every requested row.
// out[i, 0 .. F) = x[ids[i], :], out[i, F .. F + A) = att[ids[i], :]   (A = 2 * heads)
__global__ __launch_bounds__(256) void k_pack_rows2(const float *__restrict__ x, const float *__restrict__ att, const int *__restrict__ ids,
                                                    long n, int F, int A, float *__restrict__ out)
{
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const int W = F + A;
    if (t >= n * W) return;
    const long i = t / W;
    const int c = (int)(t - i * W);
    const size_t r = (size_t)ids[i];
    out[t] = c < F ? x[r * F + c] : att[r * A + (c - F)];
}

// the receiving side: in[i, :] -> x_out[i, 0 .. F), att_out[i, 0 .. A)   (the halo tails of X_ext / att_ext)
__global__ __launch_bounds__(256) void k_unpack_rows2(const float *__restrict__ in, long n, int F, int A, float *__restrict__ x_out,
                                                      float *__restrict__ att_out)
{
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const int W = F + A;
    if (t >= n * W) return;
    const long i = t / W;
    const int c = (int)(t - i * W);
    if (c < F) x_out[(size_t)i * F + c] = in[t];
    else att_out[(size_t)i * A + (c - F)] = in[t];
}

int launch_pack_rows2(const float *x, const float *att, const int *ids, int n, int feat, int att_w, float *out, void *stream_v)
{
    if (n <= 0) return GNNAGG_OK;
    const long total = (long)n * (feat + att_w);
    hipLaunchKernelGGL(k_pack_rows2, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream_v, x, att, ids, (long)n, feat,
                       att_w, out);
    HIP_TRY(hipGetLastError());
    return GNNAGG_OK;
}

int launch_unpack_rows2(const float *in, int n, int feat, int att_w, float *x_out, float *att_out, void *stream_v)
{
    if (n <= 0) return GNNAGG_OK;
    const long total = (long)n * (feat + att_w);
    hipLaunchKernelGGL(k_unpack_rows2, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream_v, in, (long)n, feat, att_w,
                       x_out, att_out);
    HIP_TRY(hipGetLastError());
    return GNNAGG_OK;
}

// ------------------------------------------------------------------ column-tiled image of X (2-D blocked mode)
// xt[t][r][0 .. tile_w) = x[r][t * tile_w ..], zero beyond feat.  Rows of x whose pitch is not a multiple of a 128-byte
// line (F = 602: 2408 B) make every 256-byte tile segment of a gather straddle three lines -- 1.5x the L2 footprint and
// traffic; the tiled image is line-aligned whatever the caller's pitch is, and costs one streaming pass over X.
// One thread per (row, 4-column quad): reads are coalesced along the row, writes are 16-byte stores.
__global__ __launch_bounds__(256) void k_tile_x(const float *__restrict__ x, float *__restrict__ xt, int rows, int feat, int tile_w,
                                                int quads_per_row)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)rows * quads_per_row;
    if (i >= total) return;
    const int r = (int)(i / quads_per_row), q = (int)(i - (long)r * quads_per_row);
    const int c = q * 4;
    const float *src = x + (size_t)r * feat + c;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c + 3 < feat) {
        if ((((uintptr_t)src) & 15) == 0) v = *reinterpret_cast<const float4 *>(src);
        else if ((((uintptr_t)src) & 7) == 0) {
            const float2 a = *reinterpret_cast<const float2 *>(src), b = *reinterpret_cast<const float2 *>(src + 2);
            v = make_float4(a.x, a.y, b.x, b.y);
        } else v = make_float4(src[0], src[1], src[2], src[3]);
    } else {
        if (c < feat) v.x = src[0];
        if (c + 1 < feat) v.y = src[1];
        if (c + 2 < feat) v.z = src[2];
    }
    const int t = c / tile_w, ct = c - t * tile_w;
    *reinterpret_cast<float4 *>(xt + ((size_t)t * rows + r) * tile_w + ct) = v;
}

int launch_tile_x(const float *x, float *xt, int rows, int feat, int tile_w, void *stream_v)
{
    if (rows <= 0) return GNNAGG_OK;
    const int ntiles = (feat + tile_w - 1) / tile_w;
    const int quads = ntiles * tile_w / 4;
    const long total = (long)rows * quads;
    hipLaunchKernelGGL(k_tile_x, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream_v, x, xt, rows, feat, tile_w, quads);
    HIP_TRY(hipGetLastError());
    return GNNAGG_OK;
}

// n 4-byte words at p = 0.  A kernel rather than hipMemsetAsync: a memset NODE of a captured HIP graph did its work on the first
// replay only (ROCm 7.2: replays 2 and 3 of the chained rows mode started from the previous replay's Yt;
// tests/test_gpu_blocked.py::test_rows_mode_on_the_blocked_order_with_the_dense_combine_behind_it).  Every fill on a path that a
// caller may capture goes through here.
__global__ __launch_bounds__(256) void k_zero_f32x4(float4 *__restrict__ p, long n4)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}
__global__ __launch_bounds__(256) void k_zero_u32(unsigned *__restrict__ p, long n)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0u;
}

int launch_zero_words(void *p, size_t n, void *stream_v)
{
    if (n == 0) return GNNAGG_OK;
    if (((uintptr_t)p & 3) != 0) return fail(GNNAGG_ERR_STATE, "internal: zero fill of an unaligned range");
    if ((n & 3) == 0 && ((uintptr_t)p & 15) == 0) {
        const long n4 = (long)(n / 4);
        hipLaunchKernelGGL(k_zero_f32x4, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream_v, reinterpret_cast<float4 *>(p), n4);
    } else {
        hipLaunchKernelGGL(k_zero_u32, dim3((unsigned)(((long)n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_v, reinterpret_cast<unsigned *>(p), (long)n);
    }
    HIP_TRY(hipGetLastError());
    return GNNAGG_OK;
}

// The inverse for the canonical rows mode on the blocked order: Yt[tile][row][tile_w] -> y[row][feat], finishing the row on the way
// (mean: / degree as finish_gcn_row does; ReLU).  One thread per (row, 4-column quad); the caller's rows may be 4-byte aligned only.
__global__ __launch_bounds__(256) void k_untile_y(const float *__restrict__ yt, float *__restrict__ y, const int *__restrict__ row_ptr,
                                                  const unsigned char *__restrict__ skip, int rows, int feat, int tile_w, int quads_per_row, int mean,
                                                  int relu, const float *__restrict__ den_t = nullptr, int ht = 1, int dhead = 1)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)rows * quads_per_row;
    if (i >= total) return;
    const int r = (int)(i / quads_per_row), q = (int)(i - (long)r * quads_per_row);
    if (skip && skip[r]) return;   // (rows another kernel writes)
    const int c = q * 4;
    const int t = c / tile_w, ct = c - t * tile_w;
    const float4 v4 = *reinterpret_cast<const float4 *>(yt + ((size_t)t * rows + r) * tile_w + ct);
    float v[4] = {v4.x, v4.y, v4.z, v4.w};
    if (den_t) {   // GAT: the softmax division (scaleArray, aggr_gat.h:207-213); a quad lies inside one head (head width % 4 == 0)
        const float d = den_t[((size_t)t * rows + r) * ht + (ht > 1 ? ct / dhead : 0)];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = d != 0.0f ? v[k] / d : 0.0f;
    }
    if (mean) {
        const float dg = (float)(row_ptr[r + 1] - row_ptr[r]);
        if (dg > 0.0f) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = v[k] / dg;
        }
    }
    if (relu) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.0f ? v[k] : 0.0f;
    }
    float *dst = y + (size_t)r * feat + c;
    if (c + 3 < feat && (((uintptr_t)dst) & 15) == 0) *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
    else if (c + 3 < feat && (((uintptr_t)dst) & 7) == 0) {
        *reinterpret_cast<float2 *>(dst) = make_float2(v[0], v[1]);
        *reinterpret_cast<float2 *>(dst + 2) = make_float2(v[2], v[3]);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (c + k < feat) dst[k] = v[k];
    }
}

int launch_untile_y(const float *yt, float *y, const int *row_ptr, const unsigned char *skip, int rows, int feat, int tile_w, int mean, int relu,
                    void *stream_v)
{
    if (rows <= 0 || feat <= 0) return GNNAGG_OK;
    const int quads = (feat + 3) / 4;
    const long total = (long)rows * quads;
    hipLaunchKernelGGL(k_untile_y, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream_v, yt, y, row_ptr, skip, rows, feat,
                       tile_w, quads, mean, relu);
    HIP_TRY(hipGetLastError());
    return GNNAGG_OK;
}

int launch_untile_y_gat(const float *yt, const float *den_t, float *y, const unsigned char *skip, int rows, int feat, int tile_w, int ht, int dhead,
                        void *stream_v)
{
    if (rows <= 0 || feat <= 0) return GNNAGG_OK;
    const int quads = (feat + 3) / 4;
    const long total = (long)rows * quads;
    hipLaunchKernelGGL(k_untile_y, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream_v, yt, y, (const int *)nullptr, skip, rows,
                       feat, tile_w, quads, 0, 0, den_t, ht, dhead);
    HIP_TRY(hipGetLastError());
    return GNNAGG_OK;
}

#ifdef GNNAGG_GEMM_TIMELINE
extern "C" __attribute__((visibility("default"))) int gnnagg_debug_set_gemm_timeline(void *d_buf)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_tl), &d_buf, sizeof(d_buf)) == hipSuccess ? 0 : 2;
}
#endif

// ------------------------------------------------------------------ compact attention terms (2-D blocked GAT)
// att is [V, H, 2] (centre term, source term interleaved per head): a tile of the span kernel needs the source terms of its
// HT heads per EDGE -- HT four-byte loads that each touch a different 64-byte att row per lane.  The compact image keeps
// them per head group hg = first head / HT as as_t[hg][v][0 .. HT) (and the centre terms as ac_t likewise): one HT * 4-byte
// load per edge, rows 16x denser in the L2 / L1 than att's.  Heads beyond H replicate head H - 1 (never stored).
__global__ __launch_bounds__(256) void k_tile_att(const float *__restrict__ att, float *__restrict__ as_t, float *__restrict__ ac_t,
                                                  int rows, int heads, int ht, int n_hg)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long per_hg = (long)rows * ht;
    if (i >= per_hg * n_hg) return;
    const int hg = (int)(i / per_hg);
    const long rem = i - (long)hg * per_hg;
    const int v = (int)(rem / ht), k = (int)(rem - (long)v * ht);
    int h = hg * ht + k;
    h = h < heads ? h : heads - 1;
    const float *cs = att + ((size_t)v * heads + h) * 2;  // (scalar loads: the caller's att may be 4-byte aligned only)
    ac_t[i] = cs[0];
    as_t[i] = cs[1];
}

int launch_tile_att(const float *att, float *as_t, float *ac_t, int rows, int heads, int ht, void *stream_v)
{
    if (rows <= 0) return GNNAGG_OK;
    const int n_hg = (heads + ht - 1) / ht;
    const long total = (long)rows * ht * n_hg;
    hipLaunchKernelGGL(k_tile_att, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream_v, att, as_t, ac_t, rows, heads,
                       ht, n_hg);
    HIP_TRY(hipGetLastError());
    return GNNAGG_OK;
}

// ------------------------------------------------------------------------- row-gather ceiling probe (measurement aid)
// What the memory system delivers to ROW GATHERS in the aggregation kernels' own access shape, by where the gathered rows live
// (one XCD's L2, the Infinity Cache, HBM): the caller chooses the window through the ids it passes.  A lane group of LANES lanes
// reads `per_group` ids (one coalesced load per LANES ids, broadcast lane to lane), gathers `active` x 16 B of row id at
// rows + id * pitch with 8 gathers in flight, XOR-consumes them and never stores (gnnagg_probe_row_gather; bench.py divides the
// bytes of its gather model by this launch's time to get the ceiling each roofline fraction is quoted against).
template <int LANES>
__global__ __launch_bounds__(256) void k_probe_row_gather(const int *__restrict__ ids, const char *__restrict__ rows, long pitch, int active,
                                                          int per_group, unsigned *sink)
{
    constexpr int U = 8, GPB = 256 / LANES;
    const int lane = threadIdx.x & (LANES - 1), grp = threadIdx.x / LANES;
    const int *my = ids + ((long)blockIdx.x * GPB + grp) * per_group;
    const char *col = rows + lane * 16;
    const bool on = lane < active;
    uint4 acc = {0, 0, 0, 0};
    int cur = my[lane];
    for (int cb = 0; cb < per_group; cb += LANES) {
        int nxt = 0;
        if (cb + LANES < per_group) nxt = my[cb + LANES + lane];
#pragma unroll 1
        for (int j = 0; j < LANES; j += U) {
            uint4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int s = __shfl(cur, j + u, LANES);
                v[u] = on ? *reinterpret_cast<const uint4 *>(col + (long)s * pitch) : uint4{0, 0, 0, 0};
            }
#pragma unroll
            for (int u = 0; u < U; ++u) { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
        }
        cur = nxt;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9U) sink[0] = acc.x;  // practically never: keeps the loads alive
}

int launch_probe_row_gather(const void *rows, long pitch, int seg_bytes, const int *ids, long n_ids, int per_group, void *stream_v)
{
    if (!rows || !ids || pitch < 16 || (pitch & 15) || seg_bytes < 16 || seg_bytes > 1024 || (seg_bytes & 15) || n_ids <= 0)
        return fail(GNNAGG_ERR_ARG, "probe_row_gather: rows / ids null, or pitch / seg_bytes not multiples of 16 (seg_bytes in 16 .. 1024)");
    const int active = seg_bytes / 16;
    int lanes = 8;
    while (lanes < active) lanes <<= 1;
    const int gpb = 256 / lanes;
    if (per_group <= 0 || per_group % lanes || n_ids % ((long)per_group * gpb))
        return fail(GNNAGG_ERR_ARG, "probe_row_gather: ids_per_group must be a multiple of the lane-group width and n_ids a multiple of ids_per_group x groups per 256-thread block");
    unsigned *sink = device_probe_sink();
    if (!sink) return fail(GNNAGG_ERR_HIP, "probe: no sink");
    const long nb = n_ids / ((long)per_group * gpb);
    if (nb > 0x7fffffffL) return fail(GNNAGG_ERR_ARG, "probe_row_gather: too many ids for one launch");
    hipStream_t stream = (hipStream_t)stream_v;
    const char *r = (const char *)rows;
    switch (lanes) {
        case 8:  hipLaunchKernelGGL((k_probe_row_gather<8>), dim3((unsigned)nb), dim3(256), 0, stream, ids, r, pitch, active, per_group, sink); break;
        case 16: hipLaunchKernelGGL((k_probe_row_gather<16>), dim3((unsigned)nb), dim3(256), 0, stream, ids, r, pitch, active, per_group, sink); break;
        case 32: hipLaunchKernelGGL((k_probe_row_gather<32>), dim3((unsigned)nb), dim3(256), 0, stream, ids, r, pitch, active, per_group, sink); break;
        default: hipLaunchKernelGGL((k_probe_row_gather<64>), dim3((unsigned)nb), dim3(256), 0, stream, ids, r, pitch, active, per_group, sink); break;
    }
    HIP_TRY(hipGetLastError());
    return GNNAGG_OK;
}

}  // namespace gnnagg
