// Microbenchmark: cycles per step of a dependent fp32 FMA chain (one wavefront), with operands from registers and from LDS.
// hipcc --offload-arch=gfx950 -O3 scripts/micro/fma_chain.hip -o /tmp/fma_chain && /tmp/fma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_reg(float *out, long long *cyc, int n, float w)
{
    float acc = threadIdx.x, x = 1.0f + threadIdx.x * 1e-7f;
    const long long t0 = __builtin_readcyclecounter();
    const unsigned long long w0 = wall_clock64();
    for (int i = 0; i < n; i += 32) {
#pragma unroll
        for (int u = 0; u < 32; ++u) acc = __builtin_fmaf(x, w, acc);
    }
    const long long t1 = __builtin_readcyclecounter();
    const unsigned long long w1 = wall_clock64();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = (long long)(w1 - w0); }
}
__global__ void k_lds(float *out, long long *cyc, int n)
{
    __shared__ float4 xs[32 * 64], ws[64];
    for (int i = threadIdx.x; i < 32 * 64; i += blockDim.x) xs[i] = make_float4(1.f, 1.0001f, 0.9999f, 1.f);
    if (threadIdx.x < 64) ws[threadIdx.x] = make_float4(1e-3f, 2e-3f, 1e-3f, 2e-3f);
    __syncthreads();
    float acc = threadIdx.x;
    const int c = threadIdx.x & 31;
    const long long t0 = __builtin_readcyclecounter();
    const unsigned long long w0 = wall_clock64();
    for (int i = 0; i < n; i += 32) {
        float4 a[8], b[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { a[q] = xs[((i / 4 + q) & 63) * 32 + c]; b[q] = ws[(i / 4 + q) & 63]; }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            acc = __builtin_fmaf(a[q].x, b[q].x, acc); acc = __builtin_fmaf(a[q].y, b[q].y, acc);
            acc = __builtin_fmaf(a[q].z, b[q].z, acc); acc = __builtin_fmaf(a[q].w, b[q].w, acc);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    const unsigned long long w1 = wall_clock64();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = (long long)(w1 - w0); }
}
int main()
{
    float *out; long long *cyc, h[2];
    hipMalloc(&out, 4096); hipMalloc(&cyc, 16);
    const int n = 1 << 20;
    int rate = 0;
    hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
    for (int rep = 0; rep < 2; ++rep) {
        k_reg<<<1, 64>>>(out, cyc, n, 1e-3f); hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
        printf("register operands: %.2f shader cycles/step, %.2f ns/step (wall clock %d kHz)\n", (double)h[0] / n, (double)h[1] / rate * 1e6 / n, rate);
        k_lds<<<1, 64>>>(out, cyc, n); hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
        printf("LDS operands (b128): %.2f shader cycles/step, %.2f ns/step\n", (double)h[0] / n, (double)h[1] / rate * 1e6 / n);
    }
    return 0;
}
