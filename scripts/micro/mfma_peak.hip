// What the matrix pipe delivers for fp32 MFMAs with NO memory traffic: independent accumulators, operands in registers.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form scripts/micro/mfma_peak.hip -o scripts/micro/mfma_peak.out
// Cases: v_mfma_f32_32x32x2_f32 with 4 accumulators (the wide GEMM's shape) and 2 (what the compiler made of the strip kernel's
// guarded loop), v_mfma_f32_16x16x4_f32 with 8; 1 / 2 / 3 wavefronts per SIMD.  Prints TFLOP/s and the implied pipe occupancy at the
// clock the run had (s_memrealtime is constant-rate, so the clock is inferred from a dependent v_add chain timed beside it).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma32(float *out, int iters, float a0, float b0)
{
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) out[0] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma16(float *out, int iters, float a0, float b0)
{
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 4; ++r) acc[i][r] = 0.0f;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 4; ++r) s += acc[i][r];
    if (s == 12345.678f) out[0] = s;
}

int main()
{
    float *out;
    CK(hipMalloc(&out, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int cus = 0;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const int iters = 4000;
    auto run = [&](const char *name, auto kern, int wg_per_cu, double flops_per_mfma, int nacc) {
        const int grid = cus * wg_per_cu;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, 200, 1.0f, 2.0f);
        CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 2.0f);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        const double mfmas = (double)grid * 4 * iters * 16 * nacc;
        printf("%-34s %d wavefront(s) per SIMD: %7.1f TFLOP/s  (%.1f ns per MFMA per SIMD)\n", name, wg_per_cu, mfmas * flops_per_mfma / best / 1e9,
               best * 1e6 / ((double)iters * 16 * nacc * wg_per_cu));
    };
    for (int w = 1; w <= 3; ++w) {
        run("v_mfma_f32_32x32x2_f32, 4 acc", k_mfma32<4>, w, 4096.0, 4);
        run("v_mfma_f32_32x32x2_f32, 2 acc", k_mfma32<2>, w, 4096.0, 2);
        run("v_mfma_f32_32x32x2_f32, 1 acc", k_mfma32<1>, w, 4096.0, 1);
        run("v_mfma_f32_16x16x4_f32, 8 acc", k_mfma16<8>, w, 2048.0, 8);
        run("v_mfma_f32_16x16x4_f32, 2 acc", k_mfma16<2>, w, 2048.0, 2);
    }
    return 0;
}
