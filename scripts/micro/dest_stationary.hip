// Microbenchmark for the DESTINATION-STATIONARY form of the 2-D blocked order: does a persistent
// kernel whose workgroups keep their output rows in LDS (64 KB per workgroup => 16 waves per CU) still reach the L2 gather
// rate, and does a soft per-XCD phase counter keep the workgroups of an XCD on the same source range?
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/dest_stationary.hip -o scripts/micro/dest_stationary.out
// Model of the real kernel: 8 XCDs x WPX persistent workgroups; every workgroup sweeps NPH phases; in phase q its lane
// groups (16 lanes x float4 = one 256-byte tile row per gather) walk a span of ~per_span edges whose sources are drawn
// uniformly from window q % P (a `slice`-byte piece of the tile image); every `glen` edges the accumulators are added into
// an LDS row.  Word format of the id stream: bit 31 = last edge of its group, bits 30..21 = LDS row, bits 20..0 = row inside
// the range.  jitter: the work of (workgroup, phase) varies by +-jit % (what makes unsynchronised workgroups drift apart).
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

__host__ __device__ __forceinline__ int span_len(int b, int q, int per_span, int jit)
{
    if (jit == 0) return per_span;
    const int r = (int)(hash32((unsigned)(b * 7919 + q) * 2654435761U) % (unsigned)(2 * jit + 1)) - jit;  // [-jit, jit] %
    int n = per_span + per_span * r / 100;
    return (n + 15) & ~15;
}

// ids[((b * nph + q) * LG + g) * span_cap + i]
__global__ void k_make_ids(unsigned *ids, int nwg, int nph, int LG, int span_cap, int per_span, int jit, int window, int glen, int rows_lds)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)nwg * nph * LG * span_cap;
    if (i >= total) return;
    const int e = (int)(i % span_cap);
    const long sp = i / span_cap;
    const int g = (int)(sp % LG);
    const long bq = sp / LG;
    const int q = (int)(bq % nph), b = (int)(bq / nph);
    const int n = span_len(b, q, per_span, jit);
    unsigned w = hash32((unsigned)i * 2654435761U + 12345u) % (unsigned)window;
    const int grp = e / glen;
    const unsigned lrow = (unsigned)((g * 131 + grp * 17 + q) % rows_lds);
    w |= lrow << 21;
    if ((e % glen) == glen - 1 || e == n - 1) w |= 0x80000000u;
    ids[i] = w;
}

template <int SRC>
__device__ __forceinline__ unsigned row_bcast(unsigned v)
{
    return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x150 + SRC, 0xf, 0xf, true);
}

template <int N, class Fn>
__device__ __forceinline__ void static_for(Fn &&f)
{
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

struct Args {
    const unsigned *ids;
    const float *x;       // [P][window][64]
    unsigned *cnt;        // [8][nph]
    float *y;             // [nwg][rows_lds][64]
    int nph, P, window, per_span, span_cap, jit, rows_lds, wpx, slack, spin_limit, mode, samewin;
    unsigned *sink;
};

// one batch of 8 gathers of the window held in `s` (lanes J..J+7)
template <int J>
__device__ __forceinline__ void issue8(float4 (&v)[8], unsigned s, const char *xr, unsigned lane_boff)
{
    static_for<8>([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        const unsigned sid = row_bcast<J + u>(s);
        v[u] = *reinterpret_cast<const float4 *>(xr + (((sid & 0x1fffffu) << 8) | lane_boff));
    });
}

template <int J>
__device__ __forceinline__ void consume8(const float4 (&v)[8], unsigned s, float4 &acc, float *lds, int lane)
{
    static_for<8>([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        const unsigned sid = row_bcast<J + u>(s);
        acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w;
        if (sid & 0x80000000u) {  // lane-group uniform
            float4 *p = reinterpret_cast<float4 *>(lds + ((sid >> 21) & 0x3ffu) * 64 + lane * 4);
            float4 t = *p;
            t.x += acc.x; t.y += acc.y; t.z += acc.z; t.w += acc.w;
            *p = t;
            acc = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    });
}

// mode 0: no barrier at all (lane groups free-run: the upper bound; LDS adds of a row are then unordered across phases)
// mode 1: workgroup barrier at every phase boundary, nothing else
// mode 2: + blocking flow control (atomic add, then spin on the counter of phase q - slack)
// mode 3: + non-blocking form: fire-and-forget add; the counter of phase q - 1 - slack is requested at the START of phase q and
//         only re-read (spin) if that early value says "not yet"
// PFID: the id windows stream across the phase boundary (next phase's first two windows requested before the barrier)
template <int NT, bool PFID>
__global__ __launch_bounds__(NT) void k_ds(const Args a)
{
    extern __shared__ float lds[];
    __shared__ unsigned early;
    constexpr int LG = NT / 16;
    const int lane = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int b = blockIdx.x, xcd = b & 7;
    for (int i = threadIdx.x; i < a.rows_lds * 16; i += NT) reinterpret_cast<float4 *>(lds)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    const unsigned lane_boff = (unsigned)lane * 16u;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const unsigned *my = a.ids + (((long)b * a.nph + 0) * LG + g) * a.span_cap;
    unsigned cur = my[lane];
    unsigned nxt = my[16 + lane];
    for (int q = 0; q < a.nph; ++q) {
        const int n = span_len(b, q, a.per_span, a.jit);
        const unsigned *my_next = a.ids + (((long)b * a.nph + (q + 1 < a.nph ? q + 1 : q)) * LG + g) * a.span_cap;
        const char *xr = reinterpret_cast<const char *>(a.x) + (size_t)(a.samewin ? 0 : q % a.P) * a.window * 256;
        unsigned early_v = 0;
        const int wq = q - 1 - a.slack;
        if (a.mode == 3 && threadIdx.x == 0 && wq >= 0)
            early_v = __hip_atomic_load(&a.cnt[xcd * a.nph + wq], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!PFID) { cur = my[lane]; nxt = 16 < n ? my[16 + lane] : 0u; }
        for (int cb = 0; cb < n; cb += 16) {
            unsigned nx2;
            if (PFID) nx2 = cb + 32 < n ? my[cb + 32 + lane] : my_next[cb + 32 - n + lane];
            else nx2 = cb + 32 < n ? my[cb + 32 + lane] : 0u;
            float4 A[8];
            issue8<0>(A, cur, xr, lane_boff);
            consume8<0>(A, cur, acc, lds, lane);
            issue8<8>(A, cur, xr, lane_boff);
            consume8<8>(A, cur, acc, lds, lane);
            cur = nxt;
            nxt = nx2;
        }
        my = my_next;
        // ---- phase boundary
        if (a.mode == 0) continue;
        if (a.mode == 1) { __syncthreads(); continue; }
        if (a.mode == 2) {
            __syncthreads();
            if (threadIdx.x == 0) {
                __hip_atomic_fetch_add(&a.cnt[xcd * a.nph + q], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int w = q - a.slack;
                if (w >= 0) {
                    int it = 0;
                    while (__hip_atomic_load(&a.cnt[xcd * a.nph + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)a.wpx && it < a.spin_limit) {
                        __builtin_amdgcn_s_sleep(8);
                        ++it;
                    }
                }
            }
            __syncthreads();
            continue;
        }
        // mode 3
        if (threadIdx.x == 0) {
            // this workgroup may start phase q + 1 once every workgroup of the XCD has finished phase q - slack... the early
            // value (requested a whole phase ago) was for phase q - 1 - slack; the fresh test below is only reached when the
            // early one fails
            if (wq >= 0 && early_v < (unsigned)a.wpx) {
                int it = 0;
                while (__hip_atomic_load(&a.cnt[xcd * a.nph + wq], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)a.wpx && it < a.spin_limit) {
                    __builtin_amdgcn_s_sleep(8);
                    ++it;
                }
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(&a.cnt[xcd * a.nph + q], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // result unused: no wait
    }
    __syncthreads();
    for (int i = threadIdx.x; i < a.rows_lds * 16; i += NT)
        reinterpret_cast<float4 *>(a.y + (size_t)b * a.rows_lds * 64)[i] = reinterpret_cast<float4 *>(lds)[i];
}

int main(int argc, char **argv)
{
    int dev_cus = 256;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0)); dev_cus = prop.multiProcessorCount;
    printf("device: %s, %d CUs\n", prop.name, dev_cus);
    const int P = 15;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct Case { const char *name; int nt; int pfid; int wg_per_cu; int rows_lds; long slice; int per_span; int jit; int slack; int waves; int glen; int mode; int samewin; };
    std::vector<Case> cases;
    cases.push_back({"1 range free-running   ", 512, 0, 2, 256, 2 << 20, 256, 0, 0, 60, 32, 0, 1});
    // what the real reddit-shaped case offers: (rows per lane group) x (edges per (row, range)) edges per span and phase
    for (int jit : {0, 20}) {
        cases.push_back({"sweep 2x512thr span 128", 512, 0, 2, 256, 2L << 20, 128, jit, 1, 2, 16, 3, 0});
        cases.push_back({"sweep 2x512thr span 256", 512, 0, 2, 256, 2L << 20, 256, jit, 1, 2, 16, 3, 0});
        cases.push_back({"sweep 2x256thr span 256", 256, 0, 2, 256, 2L << 20, 256, jit, 1, 2, 16, 3, 0});
        cases.push_back({"sweep 2x256thr span 512", 256, 0, 2, 256, 4L << 20, 512, jit, 1, 2, 32, 3, 0});
        cases.push_back({"sweep 2x512thr span 256 4MB", 512, 0, 2, 256, 4L << 20, 256, jit, 1, 2, 32, 3, 0});
        cases.push_back({"sweep 1x1024.. n/a        ", 512, 0, 1, 512, 2L << 20, 256, jit, 1, 2, 16, 3, 0});
    }
    size_t ids_cap = 0;
    unsigned *ids = nullptr, *cnt = nullptr; float *x = nullptr, *y = nullptr;
    CK(hipMalloc(&x, (size_t)64 * (4 << 20)));
    CK(hipMemset(x, 0, (size_t)64 * (4 << 20)));
    CK(hipMalloc(&cnt, 8 * 4096 * sizeof(unsigned)));
    CK(hipMalloc(&y, (size_t)4096 * 1024 * 256));
    for (const Case &c : cases) {
        const int single = c.name[0] == '1';
        const int Pc = single ? 1 : (int)((60L << 20) / c.slice);
        const int nph = single ? c.waves : Pc * c.waves * (c.samewin ? 1 : 1);
        const int nwg = dev_cus * c.wg_per_cu, wpx = nwg / 8, LG = c.nt / 16;
        const int window = (int)(c.slice / 256);
        const int span_cap = ((c.per_span + c.per_span * c.jit / 100) + 15) & ~15;
        const size_t n_ids = (size_t)nwg * nph * LG * span_cap;
        if (n_ids > ids_cap) { if (ids) CK(hipFree(ids)); CK(hipMalloc(&ids, n_ids * 4)); ids_cap = n_ids; }
        k_make_ids<<<(unsigned)((n_ids + 255) / 256), 256>>>(ids, nwg, nph, LG, span_cap, c.per_span, c.jit, window, c.glen, c.rows_lds);
        CK(hipDeviceSynchronize());
        // useful bytes
        double edges = 0;
        for (int b = 0; b < nwg; ++b)
            for (int q = 0; q < nph; ++q) edges += (double)span_len(b, q, c.per_span, c.jit) * LG;
        Args a{ids, x, cnt, y, nph, Pc, window, c.per_span, span_cap, c.jit, c.rows_lds, wpx, c.slack, 20000, c.mode, c.samewin, cnt + 8 * 4096 - 1};
        const size_t lds = (size_t)c.rows_lds * 256;
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemsetAsync(cnt, 0, 8 * 4096 * sizeof(unsigned)));
            CK(hipEventRecord(e0));
#define LAUNCH(NT_, PF_) { CK(hipFuncSetAttribute((const void *)k_ds<NT_, PF_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); k_ds<NT_, PF_><<<nwg, NT_, lds>>>(a); }
            if (c.nt == 512) { if (c.pfid) LAUNCH(512, true) else LAUNCH(512, false) }
            else { if (c.pfid) LAUNCH(256, true) else LAUNCH(256, false) }
            CK(hipGetLastError());
            CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0) best = ms < best ? ms : best;
        }
        printf("%s mode %d pfid %d slice %ld MB x %2d jit %2d slack %2d span %4d: %8.1f GB/s useful  (%.3f ms, %d phases, %.1f us/phase)\n", c.name, c.mode, c.pfid,
               c.slice >> 20, Pc, c.jit, c.slack, c.per_span, edges * 256 / best / 1e6, best, nph, best * 1e3 / nph);
        fflush(stdout);
    }
    return 0;
}
