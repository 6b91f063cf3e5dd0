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
// the same chain with the weight as a DPP row_newbcast operand of the FMA (one weight register per 16 steps), from registers
#define DPP_STEP(X, I) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:" #I " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(w), "v"(X));
#define DPP_STEP16(XA, XB, XC, XD)                                                                              \
    DPP_STEP(XA.x, 0) DPP_STEP(XA.y, 1) DPP_STEP(XA.z, 2) DPP_STEP(XA.w, 3) DPP_STEP(XB.x, 4) DPP_STEP(XB.y, 5) DPP_STEP(XB.z, 6) DPP_STEP(XB.w, 7) \
    DPP_STEP(XC.x, 8) DPP_STEP(XC.y, 9) DPP_STEP(XC.z, 10) DPP_STEP(XC.w, 11) DPP_STEP(XD.x, 12) DPP_STEP(XD.y, 13) DPP_STEP(XD.z, 14) DPP_STEP(XD.w, 15)
__global__ void k_reg_dpp(float *out, long long *cyc, int n, float w0f)
{
    float acc = threadIdx.x, w = w0f * (1 + (threadIdx.x & 15));
    float4 x = make_float4(1.0f + threadIdx.x * 1e-7f, 1.0001f, 0.9999f, 1.f);
    const long long t0 = __builtin_readcyclecounter();
    const unsigned long long w0 = wall_clock64();
    for (int i = 0; i < n; i += 32) {
        DPP_STEP16(x, x, x, x)
        DPP_STEP16(x, x, x, x)
    }
    const long long t1 = __builtin_readcyclecounter();
    const unsigned long long w1 = wall_clock64();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = (long long)(w1 - w0); }
}
// LDS-fed, 16-step batches: 4 ds_read_b128 + 1 ds_read_b32, three batches in flight, fused DPP steps.  HALF: only lanes < 32 run it
template <bool HALF>
__global__ void k_lds_dpp(float *out, long long *cyc, int n)
{
    __shared__ float4 xs[32 * 64];
    __shared__ float ws[256];
    for (int i = threadIdx.x; i < 32 * 64; i += blockDim.x) xs[i] = make_float4(1.f, 1.0001f, 0.9999f, 1.f);
    for (int i = threadIdx.x; i < 256; i += blockDim.x) ws[i] = 1e-3f * (1 + (i & 3));
    __syncthreads();
    float acc = threadIdx.x;
    const int c = threadIdx.x & 31;
    if (HALF && threadIdx.x >= 32) return;
    struct B { float4 x[4]; float w; };
    auto load = [&](B &b, int k) {
#pragma unroll
        for (int q = 0; q < 4; ++q) b.x[q] = xs[((k / 4 + q) & 63) * 32 + c];
        b.w = ws[(k & 255 & ~15) + (threadIdx.x & 15)];
    };
    const long long t0 = __builtin_readcyclecounter();
    const unsigned long long w0 = wall_clock64();
    B b0, b1, b2;
    load(b0, 0); load(b1, 16); load(b2, 32);
    for (int i = 0; i < n; i += 48) {
        { float w = b0.w; DPP_STEP16(b0.x[0], b0.x[1], b0.x[2], b0.x[3]) }
        __builtin_amdgcn_sched_barrier(0); load(b0, i + 48); __builtin_amdgcn_sched_barrier(0);
        { float w = b1.w; DPP_STEP16(b1.x[0], b1.x[1], b1.x[2], b1.x[3]) }
        __builtin_amdgcn_sched_barrier(0); load(b1, i + 64); __builtin_amdgcn_sched_barrier(0);
        { float w = b2.w; DPP_STEP16(b2.x[0], b2.x[1], b2.x[2], b2.x[3]) }
        __builtin_amdgcn_sched_barrier(0); load(b2, i + 80); __builtin_amdgcn_sched_barrier(0);
    }
    const long long t1 = __builtin_readcyclecounter();
    const unsigned long long w1 = wall_clock64();
    out[threadIdx.x] = acc + b0.w + b1.w + b2.w;
    if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = (long long)(w1 - w0); }
}
// the round-3 consumer as it is in the kernel: lanes < 32 only, 2 x (8 + 8) ds_read_b128 in flight
__global__ void k_lds_half(float *out, long long *cyc, int n)
{
    __shared__ float4 xs[32 * 64], ws[64];
    for (int i = threadIdx.x; i < 32 * 64; i += blockDim.x) xs[i] = make_float4(1.f, 1.0001f, 0.9999f, 1.f);
    if (threadIdx.x < 64) ws[threadIdx.x] = make_float4(1e-3f, 2e-3f, 1e-3f, 2e-3f);
    __syncthreads();
    float acc = threadIdx.x;
    const int c = threadIdx.x & 31;
    if (threadIdx.x >= 32) return;
    const long long t0 = __builtin_readcyclecounter();
    const unsigned long long w0 = wall_clock64();
    float4 a[8], b[8], a2[8], b2[8];
    auto load = [&](float4 (&A)[8], float4 (&Bv)[8], int i) {
#pragma unroll
        for (int q = 0; q < 8; ++q) { A[q] = xs[((i / 4 + q) & 63) * 32 + c]; Bv[q] = ws[(i / 4 + q) & 63]; }
    };
    auto steps = [&](const float4 (&A)[8], const float4 (&Bv)[8]) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            acc = __builtin_fmaf(A[q].x, Bv[q].x, acc); acc = __builtin_fmaf(A[q].y, Bv[q].y, acc);
            acc = __builtin_fmaf(A[q].z, Bv[q].z, acc); acc = __builtin_fmaf(A[q].w, Bv[q].w, acc);
        }
    };
    load(a, b, 0);
    for (int i = 0; i < n; i += 64) {
        load(a2, b2, i + 32);
        steps(a, b);
        load(a, b, i + 64);
        steps(a2, b2);
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
        const int n48 = (n / 96) * 96;
        k_reg_dpp<<<1, 64>>>(out, cyc, n, 1e-3f); hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
        printf("register operands, v_fmac_f32_dpp row_newbcast: %.2f shader cycles/step, %.2f ns/step\n", (double)h[0] / n, (double)h[1] / rate * 1e6 / n);
        k_lds_dpp<false><<<1, 64>>>(out, cyc, n48); hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
        printf("LDS 16-step batches (4 b128 + 1 b32, 3 in flight), DPP weights, 64 lanes: %.2f shader cycles/step, %.2f ns/step\n", (double)h[0] / n48, (double)h[1] / rate * 1e6 / n48);
        k_lds_dpp<true><<<1, 64>>>(out, cyc, n48); hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
        printf("LDS 16-step batches, DPP weights, lanes < 32 only: %.2f shader cycles/step, %.2f ns/step\n", (double)h[0] / n48, (double)h[1] / rate * 1e6 / n48);
        k_lds_half<<<1, 64>>>(out, cyc, n); hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
        printf("LDS operands (b128) as in the round-3 kernel (lanes < 32, 2 x 16 reads in flight): %.2f shader cycles/step, %.2f ns/step\n", (double)h[0] / n, (double)h[1] / rate * 1e6 / n);
    }
    return 0;
}
