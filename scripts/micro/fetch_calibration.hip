// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on this library's OWN access patterns (guide
// MI355X_MICROARCH.md section HBM: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read ... other access
// widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern").
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/fetch_calibration.hip -o scripts/micro/fetch_calibration.out
//   rocprofv3 --pmc FETCH_SIZE -d DIR -o pmc --output-format csv -- scripts/micro/fetch_calibration.out      (program directly after --)
// Every case is ONE kernel launch of its own name with a byte count known in advance: rows are gathered UNIFORMLY from a window far
// larger than L2 + Infinity Cache (5 GB against 32 MB + 256 MB), so practically every gathered segment has to come over the fabric
// from HBM; the launch order and the ids are fixed (hash of the index).  The id path is the kernels': one coalesced load of LANES ids,
// ds_bpermute broadcast, 8 gathers in flight, XOR-consumed.  scripts/fetch_calibration_summary.py divides known bytes by the counter.
//   cal_stream16        6 GB read, 16 B per lane, grid-stride (the guide's calibrated case: expect raw FETCH_SIZE = 1/2)
//   cal_g512_p512       512-B segments of 512-B rows (F = 128: the headline kernel's gathers), line-aligned
//   cal_g256_p256       256-B segments (the 2-D blocked order's tile rows)
//   cal_g128_p128       128-B segments
//   cal_g400_p400       400-B rows at a 400-B pitch (F = 100, config P1): segments straddle 128-B lines
//   cal_g2408_p2408     NOT run (F = 602 is re-tiled to 256-B rows before it is gathered)
//   cal_g512_mall       512-B segments from an 80 MB window (Infinity-Cache resident, like config A's X): does a MALL hit count?
//   cal_write16         2 GB written, 16 B per lane (WRITE_SIZE)
//   cal_write512_rows   512-B rows written by 32-lane groups to rows in hashed order (the result stores' shape), 2 GB
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned hash32(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

__global__ void k_make_ids(int *ids, long n, unsigned window, unsigned seed)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ids[i] = (int)(hash32((unsigned)i * 2654435761U + seed) % window);
}

// LANES lanes x 16 B = one gathered segment; ACTIVE <= LANES lanes really load (400-B rows: 25 of 32)
template <int LANES, int ACTIVE, int TAG>
__global__ __launch_bounds__(256) void cal_gather(const int *__restrict__ ids, const char *__restrict__ x, long pitch, int per_group, unsigned *sink)
{
    constexpr int U = 8, GPB = 256 / LANES;
    const int lane = threadIdx.x & (LANES - 1), grp = threadIdx.x / LANES;
    const int *my = ids + ((long)blockIdx.x * GPB + grp) * per_group;
    const char *xcol = x + lane * 16;
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
                v[u] = lane < ACTIVE ? *reinterpret_cast<const uint4 *>(xcol + (long)s * pitch) : uint4{0, 0, 0, 0};
            }
#pragma unroll
            for (int u = 0; u < U; ++u) { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
        }
        cur = nxt;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9U) sink[0] = acc.x;
}

__global__ void cal_stream16(const uint4 *__restrict__ x, long n, unsigned *sink)
{
    uint4 acc = {0, 0, 0, 0};
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const uint4 v = x[i];
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9U) sink[0] = acc.x;
}

__global__ void cal_write16(uint4 *__restrict__ x, long n)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) x[i] = uint4{1u, 2u, 3u, (unsigned)i};
}

// one 32-lane group writes one 512-B row; rows visited in hashed order (a permutation: odd multiplier modulo a power of two)
__global__ __launch_bounds__(256) void cal_write512_rows(char *__restrict__ y, unsigned rows_pow2)
{
    const unsigned g = blockIdx.x * 8u + threadIdx.x / 32u;
    if (g >= rows_pow2) return;
    const unsigned r = (g * 2654435761u) & (rows_pow2 - 1u);
    *reinterpret_cast<uint4 *>(y + (size_t)r * 512 + (threadIdx.x & 31u) * 16) = uint4{g, r, 3u, 4u};
}

int main()
{
    const size_t xbytes = (size_t)6 << 30;
    char *x; int *ids; unsigned *sink;
    CK(hipMalloc(&x, xbytes)); CK(hipMemset(x, 1, xbytes)); CK(hipMalloc(&sink, 4));
    const int nblocks = 256 * 8 * 4;
    const long max_ids = (long)nblocks * 32 * 2048;
    CK(hipMalloc(&ids, max_ids * sizeof(int)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto ms = [&]() { float t; CK(hipEventElapsedTime(&t, e0, e1)); return t; };
    // name, known useful bytes, bytes in 64-B sectors, bytes in 128-B lines, ms
    auto report = [&](const char *name, double useful, double sect64, double line128, float t) {
        printf("CAL %-18s useful_bytes %.0f sector64_bytes %.0f line128_bytes %.0f ms %.3f useful_GBps %.1f\n", name, useful, sect64, line128, t, useful / t / 1e6);
    };
    {
        const long n = (long)(xbytes / 16);
        CK(hipEventRecord(e0)); cal_stream16<<<256 * 16, 256>>>((const uint4 *)x, n, sink); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        report("cal_stream16", (double)xbytes, (double)xbytes, (double)xbytes, ms());
    }
    const size_t window_bytes = (size_t)5 << 30;
#define GATHER_CASE(NAME, LANES, ACTIVE, TAG, PITCH, WINDOW_BYTES)                                                              \
    {                                                                                                                           \
        const long pitch = PITCH, seg = (long)ACTIVE * 16;                                                                       \
        const unsigned window = (unsigned)((WINDOW_BYTES) / pitch);                                                              \
        const int gpb = 256 / LANES, per_group = 2048 / (LANES / 8);                                                             \
        const long n = (long)nblocks * gpb * per_group;                                                                          \
        k_make_ids<<<(unsigned)((n + 255) / 256), 256>>>(ids, n, window, 777u + TAG);                                            \
        CK(hipDeviceSynchronize());                                                                                              \
        CK(hipEventRecord(e0)); cal_gather<LANES, ACTIVE, TAG><<<nblocks, 256>>>(ids, x, pitch, per_group, sink);                \
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());                                                                      \
        /* expected sectors / lines per segment when the row start is uniform over its 16-B-aligned offsets modulo the unit */  \
        double s64 = 0, l128 = 0;                                                                                                \
        const long period64 = 64 / 16, period128 = 128 / 16;                                                                     \
        for (long r = 0; r < 1024; ++r) {                                                                                        \
            const long b = (r * pitch) % 128;                                                                                    \
            s64 += (double)(((b % 64) + seg + 63) / 64) * 64;                                                                    \
            l128 += (double)((b + seg + 127) / 128) * 128;                                                                       \
        }                                                                                                                        \
        (void)period64; (void)period128;                                                                                         \
        report(NAME, (double)n * seg, (double)n * s64 / 1024, (double)n * l128 / 1024, ms());                                    \
    }
    GATHER_CASE("cal_g512_p512", 32, 32, 1, 512, window_bytes)
    GATHER_CASE("cal_g256_p256", 16, 16, 2, 256, window_bytes)
    GATHER_CASE("cal_g128_p128", 8, 8, 3, 128, window_bytes)
    GATHER_CASE("cal_g400_p400", 32, 25, 4, 400, window_bytes)
    GATHER_CASE("cal_g512_mall", 32, 32, 5, 512, (size_t)80 << 20)
    {
        const long n = (long)(((size_t)2 << 30) / 16);
        CK(hipEventRecord(e0)); cal_write16<<<256 * 16, 256>>>((uint4 *)x, n); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        report("cal_write16", (double)n * 16, (double)n * 16, (double)n * 16, ms());
    }
    {
        const unsigned rows = 1u << 22;   // 4 M rows x 512 B = 2 GB
        CK(hipEventRecord(e0)); cal_write512_rows<<<rows / 8, 256>>>(x, rows); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        report("cal_write512_rows", (double)rows * 512, (double)rows * 512, (double)rows * 512, ms());
    }
    return 0;
}
