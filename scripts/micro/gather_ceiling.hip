// Microbenchmark: what MI355X's memory system delivers to ROW GATHERS (the access pattern of the aggregation kernels),
// by where the gathered rows live: one XCD's L2 (4 MB), the Infinity Cache (256 MB) or HBM.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/gather_ceiling.hip -o scripts/micro/gather_ceiling.out
//   scripts/micro/gather_ceiling.out            (prints one line per case; GB/s = useful gathered bytes / time)
// A lane group of LANES lanes gathers LANES*16 contiguous bytes of row id (at byte offset `col_off` of a row of `pitch`
// bytes); ids arrive like the kernels' neighbor ids: one coalesced load per LANES ids, broadcast with ds_bpermute, U = 8
// gathers in flight; the data is consumed with XORs (no dependent FP chain) and never stored.
// Cases: window = rows the ids of ONE XCD's workgroups are drawn from (uniformly); "shared" = all XCDs draw from the same
// window (so the Infinity Cache / HBM see one copy), "private" = XCD x draws from window x.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned hash32(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

// ids[b * per_block + i]: block b (XCD b % 8) draws from [base, base + window), base = private ? (b % 8) * window : 0
__global__ void k_make_ids(int *ids, long n, int per_block, int window, int priv, unsigned seed)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int b = (int)(i / per_block);
    const unsigned r = hash32((unsigned)i * 2654435761U + seed);
    ids[i] = (priv ? (b & 7) * window : 0) + (int)(r % (unsigned)window);
}

template <int LANES>
__global__ __launch_bounds__(256) void k_gather(const int *__restrict__ ids, const char *__restrict__ x, long pitch, int col_off,
                                                int per_group, unsigned *sink)
{
    constexpr int U = 8, GPB = 256 / LANES;
    const int lane = threadIdx.x & (LANES - 1), grp = threadIdx.x / LANES;
    const int *my = ids + ((long)blockIdx.x * GPB + grp) * per_group;
    const char *xcol = x + col_off + lane * 16;
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
                v[u] = *reinterpret_cast<const uint4 *>(xcol + (long)s * pitch);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
        }
        cur = nxt;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9U) sink[0] = acc.x;  // practically never: keeps the loads alive
}

__global__ void k_stream(const uint4 *__restrict__ x, long n, unsigned *sink)
{
    uint4 acc = {0, 0, 0, 0};
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const uint4 v = x[i];
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9U) sink[0] = acc.x;
}

static float time_ms(hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }

int main(int argc, char **argv)
{
    const size_t xbytes = (size_t)6 << 30;  // 6 GB of rows: larger than the Infinity Cache by 24x
    char *x; int *ids; unsigned *sink;
    CK(hipMalloc(&x, xbytes)); CK(hipMemset(x, 1, xbytes)); CK(hipMalloc(&sink, 4));
    const int nblocks = 256 * 8 * 4;   // 4 waves of 8 workgroups per CU
    const long max_ids = (long)nblocks * 32 * 2048;
    CK(hipMalloc(&ids, max_ids * sizeof(int)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    {
        const long n = (long)(xbytes / 16);
        k_stream<<<256 * 16, 256>>>((const uint4 *)x, n, sink);
        CK(hipEventRecord(e0)); k_stream<<<256 * 16, 256>>>((const uint4 *)x, n, sink); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        printf("stream read 6 GB (16 B/lane)                                        : %8.1f GB/s\n", xbytes / time_ms(e0, e1) / 1e6);
    }
    struct Case { const char *name; int lanes; long pitch; long window_bytes; int priv; };
    std::vector<Case> cases;
    for (int lanes : {8, 16, 32}) {
        const long seg = lanes * 16;
        for (long pitch : {seg, (long)2408}) {
            if (pitch < seg) continue;
            cases.push_back({"L2 private 1 MB ", lanes, pitch, 1 << 20, 1});
            cases.push_back({"L2 private 2 MB ", lanes, pitch, 2 << 20, 1});
            cases.push_back({"L2 private 3 MB ", lanes, pitch, 3 << 20, 1});
            cases.push_back({"L2 private 4 MB ", lanes, pitch, 4 << 20, 1});
            cases.push_back({"L2 shared 2 MB  ", lanes, pitch, 2 << 20, 0});
            cases.push_back({"MALL shared 80MB", lanes, pitch, 80 << 20, 0});
            cases.push_back({"MALL shr 200 MB ", lanes, pitch, 200 << 20, 0});
            cases.push_back({"HBM shared 5 GB ", lanes, pitch, (long)5 << 30, 0});
        }
    }
    for (const Case &c : cases) {
        const long seg = c.lanes * 16;
        // window_bytes = cache footprint of the window = rows * (128-B lines a segment touches: seg aligned -> seg/128, else +1)
        const long foot = c.pitch % 128 == 0 ? seg : seg + 128;
        long window = c.window_bytes / foot;
        if ((c.priv ? 8 : 1) * window * c.pitch + 4096 > (long)xbytes) window = ((long)xbytes - 4096) / c.pitch / (c.priv ? 8 : 1);
        const int gpb = 256 / c.lanes;
        const int per_group = 2048 / (c.lanes / 8);  // same bytes per block in every case: 8 MB
        const long n = (long)nblocks * gpb * per_group;
        k_make_ids<<<(unsigned)((n + 255) / 256), 256>>>(ids, n, gpb * per_group, (int)window, c.priv, 12345u);
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0));
            if (c.lanes == 8) k_gather<8><<<nblocks, 256>>>(ids, x, c.pitch, 0, per_group, sink);
            else if (c.lanes == 16) k_gather<16><<<nblocks, 256>>>(ids, x, c.pitch, 0, per_group, sink);
            else k_gather<32><<<nblocks, 256>>>(ids, x, c.pitch, 0, per_group, sink);
            CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
            if (rep > 0) best = time_ms(e0, e1) < best ? time_ms(e0, e1) : best;
        }
        const double bytes = (double)n * seg;
        printf("%s seg %4ld B pitch %5ld B window %8ld rows (%6.1f MB lines): %8.1f GB/s useful, %8.1f GB/s in lines, %7.3f ms\n", c.name, seg,
               c.pitch, window, window * foot / 1048576.0, bytes / best / 1e6, (double)n * foot / best / 1e6, best);
    }
    return 0;
}
