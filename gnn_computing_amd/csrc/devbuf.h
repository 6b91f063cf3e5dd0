// devbuf.h -- owning device arrays and the HIP error macro shared by the HIP translation units of libgnnagg.so.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <utility>
#include <vector>

#include "common.h"

namespace gnnagg {

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess)                                                              \
            return fail(GNNAGG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

template <class T>
struct DevBuf {  // owning device array
    T *p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    void swap(DevBuf &o) { std::swap(p, o.p); std::swap(n, o.n); }
    int upload(const std::vector<T> &h)
    {
        release();
        n = h.size();
        if (n == 0) return GNNAGG_OK;
        HIP_TRY(hipMalloc((void **)&p, n * sizeof(T)));
        HIP_TRY(hipMemcpy(p, h.data(), n * sizeof(T), hipMemcpyHostToDevice));
        return GNNAGG_OK;
    }
    int reserve(size_t want)  // grow-only scratch
    {
        if (want <= n) return GNNAGG_OK;
        release();
        HIP_TRY(hipMalloc((void **)&p, want * sizeof(T)));
        n = want;
        return GNNAGG_OK;
    }
    int alloc(size_t count)  // exactly `count` elements (at least one byte is allocated so p is never null for count 0 callers that index)
    {
        release();
        HIP_TRY(hipMalloc((void **)&p, (count ? count : 1) * sizeof(T)));
        n = count;
        return GNNAGG_OK;
    }
};

// ---- plan_gpu.hip: the blocked-order plans built on the device (no per-edge array ever crosses PCIe)
struct GpuBlockedPlan {          // localityNeighborGrouping(par_num, ng) as the segmented-stream kernel wants it
    int par_num = 0, total_cols = 0, ng = 0, G = 0, n_edges = 0, n_spans = 0, n_crows = 0, n_empty = 0;
    DevBuf<int> ptr_s, target, eperm, idx_f, span_g, rg_ptr, rg_idx, crows, empty_rows;
    std::vector<int> h_ptr_s, h_target, h_empty;
    std::vector<long> span_cost_prefix;
};
struct GpuChainPlan {            // locality_schedule (one group per (row, range)) for the chained rows mode
    bool sorted_rows = false;    // false: some row lists its neighbors out of order -> not this path (nothing else is filled)
    int par_num = 0, total_cols = 0, G = 0, n_edges = 0, n_hub = 0;
    DevBuf<int> ptr_s, target, eperm, idx_f, span_g, r1;
    DevBuf<unsigned char> hub_mask;
    std::vector<int> h_ptr_s, h_target, span0;
    std::vector<std::vector<long>> cost;
};
int gpu_max_col(const int *d_idx, int E, hipStream_t stream, int *max_col);
int gpu_build_blocked_plan(const int *d_ptr, const int *d_idx, int V, int E, int par_num, int total_cols, int ng, int span_edges,
                           hipStream_t stream, GpuBlockedPlan &out);
int gpu_build_chain_plan(const int *d_ptr, const int *d_idx, const int *h_ptr, int V, int E, int par_num, int total_cols, int hub_edges,
                         int span_edges, hipStream_t stream, GpuChainPlan &out);

}  // namespace gnnagg
