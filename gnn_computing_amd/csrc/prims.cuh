// prims.cuh -- the three device-wide primitives the plan builder needs, hand-written for wave64: a scan (any associative op, inclusive or
// exclusive), a reduction, and a STABLE least-significant-digit radix sort of (key, value) pairs by the low `bits` of the key.
// (Round 4 first used hipCUB for them: rocPRIM's kernels and their mangled names were 7 of the library's 12 MB and 1.5 of its 4 build
// minutes.  The plan needs one sort by a <= 10-bit key over the edges, two short ones over the groups and four scans: HBM-bound passes.)
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>

#include "devbuf.h"

namespace gnnagg {
namespace prims {

constexpr int kBlock = 256, kItems = 16, kTile = kBlock * kItems;   // elements per workgroup and pass

struct OpMax { static constexpr int identity = INT_MIN; __device__ static int apply(int a, int b) { return a > b ? a : b; } };
struct OpSum { static constexpr int identity = 0;       __device__ static int apply(int a, int b) { return a + b; } };

// inclusive scan of the 256 per-thread values of a workgroup; returns this thread's EXCLUSIVE prefix, *total = the workgroup's total
template <class OP>
__device__ __forceinline__ int block_exclusive(int v, int *lds /* [4] */, int *total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d, 64);
        if (lane >= d) incl = OP::apply(o, incl);
    }
    if (lane == 63) lds[wave] = incl;
    __syncthreads();
    int carry = OP::identity, tot = OP::identity;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) {
        const int t = lds[w];
        if (w < wave) carry = OP::apply(carry, t);
        tot = OP::apply(tot, t);
    }
    __syncthreads();
    *total = tot;
    int excl = __shfl_up(incl, 1, 64);
    if (lane == 0) excl = OP::identity;
    return OP::apply(carry, excl);
}

// pass 1: the total of every tile
template <class OP>
__global__ __launch_bounds__(kBlock) void k_scan_tile_totals(const int *__restrict__ in, long n, int *__restrict__ tile_tot)
{
    __shared__ int lds[4];
    const long base = (long)blockIdx.x * kTile + (long)threadIdx.x * kItems;
    int v = OP::identity;
#pragma unroll
    for (int i = 0; i < kItems; ++i)
        if (base + i < n) v = OP::apply(v, in[base + i]);
    int tot;
    (void)block_exclusive<OP>(v, lds, &tot);
    if (threadIdx.x == 0) tile_tot[blockIdx.x] = tot;
}

// pass 2 (one workgroup): exclusive scan of the tile totals, in place, 256 at a time with a running carry
template <class OP>
__global__ __launch_bounds__(kBlock) void k_scan_totals(int *__restrict__ tile_tot, int nt)
{
    __shared__ int lds[4];
    int carry = OP::identity;
    for (int b = 0; b < nt; b += kBlock) {
        const int i = b + (int)threadIdx.x;
        const int v = i < nt ? tile_tot[i] : OP::identity;
        int tot;
        const int ex = block_exclusive<OP>(v, lds, &tot);
        if (i < nt) tile_tot[i] = OP::apply(carry, ex);
        carry = OP::apply(carry, tot);
    }
}

// pass 3: every tile rescanned behind its carry (in == out is fine: a thread reads its items before it writes them)
template <class OP, bool EXCLUSIVE>
__global__ __launch_bounds__(kBlock) void k_scan_apply(const int *__restrict__ in, int *__restrict__ out, long n, const int *__restrict__ tile_carry)
{
    __shared__ int lds[4];
    const long base = (long)blockIdx.x * kTile + (long)threadIdx.x * kItems;
    int x[kItems];
    int v = OP::identity;
#pragma unroll
    for (int i = 0; i < kItems; ++i) {
        x[i] = base + i < n ? in[base + i] : OP::identity;
        v = OP::apply(v, x[i]);
    }
    int tot;
    int run = OP::apply(tile_carry[blockIdx.x], block_exclusive<OP>(v, lds, &tot));
#pragma unroll
    for (int i = 0; i < kItems; ++i) {
        const int incl = OP::apply(run, x[i]);
        if (base + i < n) out[base + i] = EXCLUSIVE ? run : incl;
        run = incl;
    }
}

template <class OP, bool EXCLUSIVE>
static int scan(const int *in, int *out, long n, DevBuf<int> &tmp, hipStream_t st)
{
    if (n <= 0) return GNNAGG_OK;
    const int nt = (int)((n + kTile - 1) / kTile);
    int rc = tmp.reserve((size_t)nt);
    if (rc) return rc;
    hipLaunchKernelGGL((k_scan_tile_totals<OP>), dim3(nt), dim3(kBlock), 0, st, in, n, tmp.p);
    hipLaunchKernelGGL((k_scan_totals<OP>), dim3(1), dim3(kBlock), 0, st, tmp.p, nt);
    hipLaunchKernelGGL((k_scan_apply<OP, EXCLUSIVE>), dim3(nt), dim3(kBlock), 0, st, in, out, n, tmp.p);
    HIP_TRY(hipGetLastError());
    return GNNAGG_OK;
}

template <class OP>
__global__ __launch_bounds__(kBlock) void k_reduce(const int *__restrict__ in, long n, int *__restrict__ out)
{
    __shared__ int lds[4];
    int v = OP::identity;
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long)gridDim.x * kBlock) v = OP::apply(v, in[i]);
    int tot;
    (void)block_exclusive<OP>(v, lds, &tot);
    if (threadIdx.x == 0) {
        if (OP::identity == 0) atomicAdd(out, tot);
        else atomicMax(out, tot);
    }
}

// *out (device) = reduction of in[0 .. n); *out is initialised here
template <class OP>
static int reduce(const int *in, long n, int *out, hipStream_t st)
{
    const int init = OP::identity;
    HIP_TRY(hipMemcpyAsync(out, &init, sizeof(int), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));   // (`init` lives on this stack frame)
    if (n > 0) {
        const int grid = (int)std::min<long>((n + kBlock - 1) / kBlock, 2048);
        hipLaunchKernelGGL((k_reduce<OP>), dim3(grid), dim3(kBlock), 0, st, in, n, out);
        HIP_TRY(hipGetLastError());
    }
    return GNNAGG_OK;
}

// ---- stable radix sort, 8 bits per pass.  counts[digit][tile] (digit-major) -> exclusive sum -> every tile scatters its elements in
// order: 16 rounds of 256 consecutive elements; the rank of an element among the equal digits of its round is a wave-level match
// (8 ballots) plus the counts of the wavefronts before it; the digit's running offset lives in LDS.
__global__ __launch_bounds__(kBlock) void k_sort_hist(const unsigned *__restrict__ keys, long n, int shift, int nt, int *__restrict__ counts)
{
    __shared__ int h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const long base = (long)blockIdx.x * kTile;
#pragma unroll
    for (int j = 0; j < kItems; ++j) {
        const long i = base + (long)j * kBlock + threadIdx.x;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & 255u], 1);
    }
    __syncthreads();
    counts[(size_t)threadIdx.x * nt + blockIdx.x] = h[threadIdx.x];
}

__global__ __launch_bounds__(kBlock) void k_sort_scatter(const unsigned *__restrict__ keys, const int *__restrict__ vals, long n, int shift, int nt,
                                                         const int *__restrict__ offsets, unsigned *__restrict__ keys_out, int *__restrict__ vals_out)
{
    __shared__ int base[256];          // next output position of every digit
    __shared__ int wcount[4][256];     // this round's count of every digit per wavefront
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    base[threadIdx.x] = offsets[(size_t)threadIdx.x * nt + blockIdx.x];
    const long tile0 = (long)blockIdx.x * kTile;
    for (int j = 0; j < kItems; ++j) {
#pragma unroll
        for (int w = 0; w < 4; ++w) wcount[w][threadIdx.x] = 0;
        __syncthreads();
        const long i = tile0 + (long)j * kBlock + threadIdx.x;
        const bool valid = i < n;
        const unsigned key = valid ? keys[i] : 0u;
        const int val = valid ? vals[i] : 0;
        const unsigned dg = (key >> shift) & 255u;
        unsigned long long same = __ballot(valid);    // lanes of this wavefront with the same digit as mine
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long bal = __ballot((dg >> b) & 1u);
            same &= ((dg >> b) & 1u) ? bal : ~bal;
        }
        const int rank = __popcll(same & ((1ull << lane) - 1ull));
        if (valid && rank == 0) wcount[wave][dg] = __popcll(same);
        __syncthreads();
        if (valid) {
            int pos = base[dg] + rank;
#pragma unroll
            for (int w = 0; w < 4; ++w)
                if (w < wave) pos += wcount[w][dg];
            keys_out[pos] = key;
            vals_out[pos] = val;
        }
        __syncthreads();
        base[threadIdx.x] += wcount[0][threadIdx.x] + wcount[1][threadIdx.x] + wcount[2][threadIdx.x] + wcount[3][threadIdx.x];
        __syncthreads();
    }
}

// Sorts n pairs by the low `bits` bits of the key, stably.  The result is in (keys_out, vals_out); (keys_in, vals_in) are CLOBBERED.
static int sort_pairs(unsigned *keys_in, unsigned *keys_out, int *vals_in, int *vals_out, long n, int bits, DevBuf<int> &counts, DevBuf<int> &scan_tmp,
                      hipStream_t st)
{
    if (n <= 0) return GNNAGG_OK;
    const int nt = (int)((n + kTile - 1) / kTile);
    const int passes = bits <= 0 ? 1 : (bits + 7) / 8;
    int rc = counts.reserve((size_t)256 * nt);
    if (rc) return rc;
    unsigned *ki = keys_in, *ko = keys_out;
    int *vi = vals_in, *vo = vals_out;
    for (int p = 0; p < passes; ++p) {
        hipLaunchKernelGGL(k_sort_hist, dim3(nt), dim3(kBlock), 0, st, ki, n, 8 * p, nt, counts.p);
        if ((rc = scan<OpSum, true>(counts.p, counts.p, (long)256 * nt, scan_tmp, st))) return rc;
        hipLaunchKernelGGL(k_sort_scatter, dim3(nt), dim3(kBlock), 0, st, ki, vi, n, 8 * p, nt, counts.p, ko, vo);
        HIP_TRY(hipGetLastError());
        std::swap(ki, ko);
        std::swap(vi, vo);
    }
    if (ki != keys_out) {   // an even number of passes left the result in the input buffers
        HIP_TRY(hipMemcpyAsync(keys_out, ki, (size_t)n * sizeof(unsigned), hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(vals_out, vi, (size_t)n * sizeof(int), hipMemcpyDeviceToDevice, st));
    }
    return GNNAGG_OK;
}

}  // namespace prims
}  // namespace gnnagg
