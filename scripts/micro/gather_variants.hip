// Microbenchmark: does the INSTRUCTION FORM of an L2-resident row gather change what the L1 / address path delivers?
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/gather_variants.hip -o scripts/micro/gather_variants.out
// 16-lane groups gather 256-byte segments (the 2-D blocked order's tile rows) of uniformly drawn rows of a window that is
// private to the XCD (block b draws from window b % 8); ids by coalesced window + DPP row broadcast (as k_gat_span), U
// gathers in flight, XOR-consumed.  Variants: 64-bit global addresses, SGPR base + 32-bit offset, buffer loads with the
// cache-policy bits (sc0 / nt / sc1), 8 or 16 gathers in flight.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned hash32(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

__global__ void k_make_ids(int *ids, long n, int per_block, int window, unsigned seed)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int b = (int)(i / per_block);
    ids[i] = (b & 7) * window + (int)(hash32((unsigned)i * 2654435761U + seed) % (unsigned)window);
}

template <int N, class Fn>
__device__ __forceinline__ void static_for(Fn &&f)
{
    if constexpr (N > 0) { static_for<N - 1>(f); f(std::integral_constant<int, N - 1>{}); }
}
template <int SRC>
__device__ __forceinline__ int row_bcast(int v) { return __builtin_amdgcn_mov_dpp(v, 0x150 + SRC, 0xf, 0xf, true); }

typedef unsigned u4 __attribute__((ext_vector_type(4)));

// MODE 0: 64-bit address arithmetic; 1: uniform base + (id << 8 | lane * 16); 2..: buffer load, aux = AUX
template <int MODE, int AUX, int U>
__global__ __launch_bounds__(256) void k_gather(const int *__restrict__ ids, const char *__restrict__ x, unsigned xbytes, int per_group,
                                                unsigned *sink)
{
    constexpr int LANES = 16, GPB = 256 / LANES;
    const int lane = threadIdx.x & (LANES - 1), grp = threadIdx.x / LANES;
    const int *my = ids + ((long)blockIdx.x * GPB + grp) * per_group;
    const char *xcol = x + lane * 16;
    const unsigned lane_off = lane * 16;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(x), 0, (int)xbytes, 0x00020000);
    u4 acc = {0, 0, 0, 0};
    int cur = my[lane];
    for (int cb = 0; cb < per_group; cb += LANES) {
        int nxt = 0;
        if (cb + LANES < per_group) nxt = my[cb + LANES + lane];
        static_for<LANES / U>([&](auto bc) {
            constexpr int J = decltype(bc)::value * U;
            u4 v[U];
            static_for<U>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                const unsigned s = (unsigned)row_bcast<J + u>(cur);
                if constexpr (MODE == 0) v[u] = *reinterpret_cast<const u4 *>(xcol + (long)s * 256);
                else if constexpr (MODE == 1) v[u] = *reinterpret_cast<const u4 *>(x + ((s << 8) | lane_off));
                else v[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)((s << 8) | lane_off), 0, AUX);
            });
            static_for<U>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                acc ^= v[u];
            });
        });
        cur = nxt;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9U) sink[0] = acc.x;
}

static float time_ms(hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }

template <int MODE, int AUX, int U>
static void run(const char *name, const int *ids, const char *x, unsigned xbytes, int nblocks, int per_group, unsigned *sink, hipEvent_t e0,
                hipEvent_t e1)
{
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        k_gather<MODE, AUX, U><<<nblocks, 256>>>(ids, x, xbytes, per_group, sink);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        if (rep > 0) best = time_ms(e0, e1) < best ? time_ms(e0, e1) : best;
    }
    const double bytes = (double)nblocks * 16 * per_group * 256.0;
    printf("  %-52s %8.1f GB/s  %7.3f ms\n", name, bytes / best / 1e6, best);
}

int main()
{
    const unsigned xbytes = 1u << 30;
    char *x; int *ids; unsigned *sink;
    CK(hipMalloc(&x, xbytes)); CK(hipMemset(x, 1, xbytes)); CK(hipMalloc(&sink, 4));
    const int nblocks = 256 * 8 * 4, per_group = 1024;
    const long n = (long)nblocks * 16 * per_group;
    CK(hipMalloc(&ids, n * sizeof(int)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (long wbytes : {2L << 20, 3L << 20, 4L << 20, 64L << 20}) {
        const int window = (int)(wbytes / 256);
        if ((long)window * 8 * 256 > (long)xbytes) continue;
        k_make_ids<<<(unsigned)((n + 255) / 256), 256>>>(ids, n, 16 * per_group, window, 777u);
        printf("window %ld KB per XCD (256-B segments, 16-lane groups)\n", wbytes >> 10);
        run<0, 0, 8>("global, 64-bit address, 8 in flight", ids, x, xbytes, nblocks, per_group, sink, e0, e1);
        run<1, 0, 8>("global, base + 32-bit offset, 8 in flight", ids, x, xbytes, nblocks, per_group, sink, e0, e1);
        run<1, 0, 16>("global, base + 32-bit offset, 16 in flight", ids, x, xbytes, nblocks, per_group, sink, e0, e1);
        run<2, 0, 8>("buffer, aux 0, 8 in flight", ids, x, xbytes, nblocks, per_group, sink, e0, e1);
        run<2, 0, 16>("buffer, aux 0, 16 in flight", ids, x, xbytes, nblocks, per_group, sink, e0, e1);
        run<2, 1, 8>("buffer, sc0, 8 in flight", ids, x, xbytes, nblocks, per_group, sink, e0, e1);
        run<2, 2, 8>("buffer, nt, 8 in flight", ids, x, xbytes, nblocks, per_group, sink, e0, e1);
        run<2, 3, 8>("buffer, sc0 nt, 8 in flight", ids, x, xbytes, nblocks, per_group, sink, e0, e1);
        run<2, 16, 8>("buffer, sc1, 8 in flight", ids, x, xbytes, nblocks, per_group, sink, e0, e1);
        run<2, 17, 8>("buffer, sc0 sc1, 8 in flight", ids, x, xbytes, nblocks, per_group, sink, e0, e1);
    }
    return 0;
}
