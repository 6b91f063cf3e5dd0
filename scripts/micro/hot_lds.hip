// Microbenchmark: the hot source rows of an L2-resident slice served from LDS instead of through the texture path.
// The 2-D blocked order's span kernels are bound by the TA / L1 path (64 B per clock and CU: 256-byte tile rows gathered by 16
// lanes x float4), not by the L2 behind it.  LDS has its own 128 B per clock and CU.  If the `hot` most-referenced rows of a
// (source range, column tile) slice sit in LDS (256 rows x 256 B = 64 KB per workgroup, 2 workgroups of 512 threads per CU),
// the edges that name them never enter the TA.  On the reddit-shaped graph the 256 hottest of a 16384-row range take 17.3 % of
// its edges, the 512 hottest 21.7 %.  This models the kernel's inner loop: lane groups of 16 walk spans of ids (bit 30 = hot,
// low bits = slot or row), 16-id windows, DPP row broadcast, 8 loads in flight, a running sum per lane.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/hot_lds.hip -o scripts/micro/hot_lds.out && scripts/micro/hot_lds.out
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__host__ __device__ __forceinline__ unsigned hash32(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

// sorted: 1 = the hot edges of every 16-id window come first (what a plan could arrange inside a group whose order is free),
// 2 = the hot edges of every SPAN come first
__global__ void k_make_ids(unsigned *ids, long total, int window, int hot_rows, unsigned hot_per_64k, int per_span, int sorted)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const unsigned r = hash32((unsigned)i * 2654435761U + 12345u);
    bool hot = (hash32(r ^ 0x5bd1e995u) & 0xffffu) < hot_per_64k;
    if (sorted == 1) hot = (unsigned)(i & 15) * 4096u < hot_per_64k;                          // first k of every window
    if (sorted == 2) hot = (unsigned long long)(i % per_span) * 65536ull < (unsigned long long)hot_per_64k * per_span;
    ids[i] = hot ? (0x40000000u | (r % (unsigned)hot_rows)) : (r % (unsigned)window);
}

template <int SRC>
__device__ __forceinline__ unsigned row_bcast(unsigned v)
{
    return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x150 + SRC, 0xf, 0xf, true);
}

template <int N, class Fn>
__device__ __forceinline__ void static_for(Fn &&f)
{
    [&]<int... I>(std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, N>{});
}

// MODE 0: every id through the texture path (hot ids name rows of the slice too); 1: hot ids read LDS
template <int MODE, int NT>
__global__ __launch_bounds__(NT) void k_walk(const float *__restrict__ x, const unsigned *__restrict__ ids, float *__restrict__ out, int per_span, int hot_rows)
{
    extern __shared__ float4 lds[];
    const int lane = threadIdx.x & 15, grp = threadIdx.x >> 4;
    constexpr int LG = NT / 16;
    if (MODE == 1) {
        for (int i = threadIdx.x; i < hot_rows * 16; i += NT) lds[i] = reinterpret_cast<const float4 *>(x)[i];
        __syncthreads();
    }
    const unsigned *my = ids + ((size_t)blockIdx.x * LG + grp) * per_span;
    const char *xb = reinterpret_cast<const char *>(x);
    const unsigned lane_off = lane * 16;
    float4 acc = {0, 0, 0, 0};
    unsigned w = my[lane];
    for (int cb = 0; cb < per_span; cb += 16) {
        const unsigned nx = cb + 16 < per_span ? my[cb + 16 + lane] : 0;
        static_for<2>([&](auto bc) {
            constexpr int J = decltype(bc)::value * 8;
            float4 v[8];
            static_for<8>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                const unsigned id = row_bcast<J + u>(w);
                if (MODE == 1 && (id & 0x40000000u)) v[u] = lds[((id & 0xffffu) << 4) | lane];
                else v[u] = *reinterpret_cast<const float4 *>(xb + (((id & 0xffffffu) << 8) | lane_off));
            });
            static_for<8>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w;
            });
        });
        w = nx;
    }
    reinterpret_cast<float4 *>(out)[((size_t)blockIdx.x * LG + grp) * 16 + lane] = acc;
}

int main(int argc, char **argv)
{
    const int window = argc > 1 ? atoi(argv[1]) : 16384;     // rows of the slice (x 256 B)
    const int per_span = argc > 2 ? atoi(argv[2]) : 1536;
    const int nwg = argc > 3 ? atoi(argv[3]) : 4096;
    constexpr int NT = 512, LG = NT / 16;
    const int hot_rows = 256;
    const long total = (long)nwg * LG * per_span;
    float *x, *out;
    unsigned *ids;
    CK(hipMalloc(&x, (size_t)window * 256));
    CK(hipMalloc(&out, (size_t)nwg * LG * 256));
    CK(hipMalloc(&ids, total * 4));
    CK(hipMemset(x, 0, (size_t)window * 256));
    CK(hipFuncSetAttribute((const void *)k_walk<1, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("slice %d rows (%.1f MB), spans of %d ids, %d workgroups of %d threads, %ld ids = %.1f GB of tile rows\n", window, window * 256 / 1048576.0,
           per_span, nwg, NT, total, total * 256.0 / 1e9);
    const double fr[] = {0.0, 0.10, 0.173, 0.217, 0.30, 0.50, 1.0};
    for (int sorted = 0; sorted < 3; ++sorted)
        for (double f : fr) {
            hipLaunchKernelGGL(k_make_ids, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, ids, total, window, hot_rows, (unsigned)(f * 65536), per_span, sorted);
            float ms[2] = {0, 0};
            for (int mode = 0; mode < 2; ++mode) {
                for (int it = 0; it < 4; ++it) {
                    if (it == 1) CK(hipEventRecord(e0));
                    if (mode == 0) hipLaunchKernelGGL((k_walk<0, NT>), dim3(nwg), dim3(NT), 0, 0, x, ids, out, per_span, hot_rows);
                    else hipLaunchKernelGGL((k_walk<1, NT>), dim3(nwg), dim3(NT), 65536, 0, x, ids, out, per_span, hot_rows);
                }
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms[mode], e0, e1));
                ms[mode] /= 3;
            }
            printf("order %d  hot fraction %.3f: all through TA %.3f ms (%.1f TB/s)   hot from LDS %.3f ms (%.1f TB/s)   x%.3f\n", sorted, f, ms[0],
                   total * 256.0 / ms[0] / 1e9, ms[1], total * 256.0 / ms[1] / 1e9, ms[0] / ms[1]);
        }
    return 0;
}
