// Shared helpers of the benchmark drivers: seeded N(0,1) device arrays (the reference fills its buffers with
// curandGenerateNormal(seed 123), Figure9/main.cu:20-22,48) and per-iteration hipEvent timing.
#pragma once
#include <algorithm>
#include <random>
#include <vector>

#include "../include/compat/util.h"

// `--dump DIR` (anywhere on the command line; stripped before the reference's argParse sees it): the class shim writes the operands
// of the last call of each entry point into DIR (GNNAGG_COMPAT_DUMP, include/compat/util.h) -- the numbers behind the driver
inline void strip_dump_flag(int &argc, char **argv)
{
    int w = 1;
    for (int i = 1; i < argc; ++i) {
        if (std::string(argv[i]) == "--dump" && i + 1 < argc) {
            setenv("GNNAGG_COMPAT_DUMP", argv[++i], 1);
            continue;
        }
        argv[w++] = argv[i];
    }
    argc = w;
}

inline float *device_normal(size_t count, unsigned long long seed)
{
    std::vector<float> h(count);
    std::mt19937_64 gen(seed);
    std::normal_distribution<float> d(0.f, 1.f);
    for (auto &v : h) v = d(gen);
    float *p = nullptr;
    checkHipErrors(hipMalloc2((void **)&p, sizeof(float) * std::max<size_t>(count, 1)));
    checkHipErrors(hipMemcpy(p, h.data(), sizeof(float) * count, hipMemcpyHostToDevice));
    return p;
}

// runs fn() `times` times, each bracketed by hipEvents on the default stream; returns the median in seconds
template <class Fn>
inline double median_time(int times, Fn fn)
{
    std::vector<float> ms(times);
    hipEvent_t a, b;
    checkHipErrors(hipEventCreate(&a));
    checkHipErrors(hipEventCreate(&b));
    for (int i = 0; i < times; ++i) {
        checkHipErrors(hipEventRecord(a, nullptr));
        fn();
        checkHipErrors(hipEventRecord(b, nullptr));
        checkHipErrors(hipEventSynchronize(b));
        checkHipErrors(hipEventElapsedTime(&ms[i], a, b));
    }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    std::sort(ms.begin(), ms.end());
    return ms[times / 2] * 1e-3;
}

inline void report(const char *what, double sec)
{
    const double bytes = (double)m * (4.0 * feature_len + 8.0) + (double)n * 4.0 * feature_len + 4.0 * (n + 1);
    fprintf(stderr, "{\"variant\": \"%s\", \"seconds\": %.9f, \"edges_per_s\": %.6e, \"gflops\": %.3f, \"algorithmic_gbps\": %.1f}\n",
            what, sec, (double)m / sec, getFLOP(sec), bytes / sec / 1e9);
}
