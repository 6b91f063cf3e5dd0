// Microbenchmark: issue cost of v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32 in 1, 2 and 4 independent accumulator chains
// (one wavefront; and 4 wavefronts per SIMD for the pipe's throughput).
// hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_chain.hip -o /tmp/mfma_chain && /tmp/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16 __attribute__((ext_vector_type(16)));
template <int CH>
__global__ void k16(float *out, long long *cyc, int n, float a, float b)
{
    f4 acc[CH];
    for (int c = 0; c < CH; ++c) acc[c] = {0.f, 0.f, 0.f, 0.f};
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; i += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int CH>
__global__ void k32(float *out, long long *cyc, int n, float a, float b)
{
    f16 acc[CH];
    for (int c = 0; c < CH; ++c)
        for (int j = 0; j < 16; ++j) acc[c][j] = 0.f;
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; i += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][15];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main()
{
    float *out; long long *cyc, h;
    (void)hipMalloc(&out, 1 << 20); (void)hipMalloc(&cyc, 16);
    const int n = 1 << 16;
#define RUN(K, CH, BLK, NAME)                                                                     \
    K<CH><<<1, BLK>>>(out, cyc, n, 1.0f, 1e-3f); (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); \
    printf("%-14s chains=%d waves/SIMD=%d: %.1f cycles per MFMA of one wave (%.1f per chain step)\n", NAME, CH, BLK / 256 ? BLK / 256 : 1, (double)h / n / CH, (double)h / n);
    RUN(k16, 1, 64, "16x16x4") RUN(k16, 2, 64, "16x16x4") RUN(k16, 4, 64, "16x16x4")
    RUN(k16, 1, 256, "16x16x4 4w/CU") RUN(k16, 1, 1024, "16x16x4 16w/CU")
    RUN(k32, 1, 64, "32x32x2") RUN(k32, 2, 64, "32x32x2") RUN(k32, 1, 1024, "32x32x2 16w/CU")
    return 0;
}
