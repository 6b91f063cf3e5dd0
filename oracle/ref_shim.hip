// oracle/ref_shim.hip -- C entry points over the REFERENCE's own aggregation path, for oracle/_ref/libref_gnn.so.
//
// TEST INFRASTRUCTURE ONLY (tests/ pin the CPU oracle and the product against it; nothing under gnn_computing_amd/ loads it).
//
// This file is this repo's code; everything it calls is the reference's: oracle/ref_build.sh translates the reference's
// sources WHERE THEY LIE (/root/reference/include/*.h, src/data.cu, src/util.cu) with ROCm's hipify-perl into a scratch
// directory outside the repo, compiles this shim against them with hipcc for gfx950 and deletes the scratch directory.
// No reference source is copied into the repo; only the built library lands in oracle/_ref/ (git-ignored).
//
// What runs unchanged on a 64-wide wavefront and is therefore exposed here:
//   Aggregator_GCN::run  -> aggr_gcn        (include/aggr_gcn.h:5-36,  `__shfl(x, j, 32)`: explicit 32-lane width)
//                        -> aggr_gcn_target (include/aggr_gcn.h:78-114, shared memory per 32-lane group + atomicAdd)
//   Aggregator_GAT::run  -> aggr_gat        (include/aggr_gat.h:116-164), aggr_gat_fine + scaleArray (:167-213)
//   Aggregator::schedule -> neighbor_grouping_schedule / locality_schedule / localityNeighborGrouping (graph_schedule.h:17-243)
//   Aggregator::csr2edgelist (aggregator.h:11-23,115-122), load_graph / reorderCSR (src/data.cu:4-139)
//   Aggregator_GAT::run_att / run_u_add_v / run_add_to_center / run_div_each -> attGat, u_add_v, add_to_center, each_div
//       (aggr_gat.h:5-92,395-425).  attGat and add_to_center reduce with `__shfl_down_sync(mask, v, i)` at the DEFAULT width,
//       i.e. over the hardware warp -- 64 lanes here.  They are therefore called with BLOCK_SIZE = 32 (a parameter of the
//       reference's own methods: one 32-lane warp per workgroup, so every wavefront carries exactly one of the reference's
//       warps in its lower half): lane 0 then receives v[0] + v[16], + v[8] + v[24], ... -- the reference's tree, none of the
//       inactive upper lanes on its path -- and the broadcast from lane 0 is the warp's lane 0.  With the drivers' BLOCK_SIZE
//       (128: two rows per wavefront) the reductions would mix neighbouring rows on this hardware.
//   Aggregator_GCN::runEdgeWise / run_with_nn (aggr_gcn.h:291-359,446-499), spmm<L> / valid / validReordered (spmm.h:11-91,223-265)
// NOT exposed: the backward kernel (aggr_gat.h:222-296, "Experiment", no caller): the product implements the mathematics its
// comments describe, not the code, so there is nothing to compare.
// (the library headers first: aggregator.h:5-6 turns __shfl / __shfl_down into macros, which must not be in force when
// HIP's own headers declare functions of those names)
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hipsparse.h>
#include <hipblas.h>
#include <hiprand.h>
#include <unistd.h>

#include <cstring>
#include <string>
#include <vector>

#include "aggr_gcn.h"
#include "aggr_gat.h"
#include "spmm.h"

// dense.h:16-22 hands cublasSgeam a NULL beta and a NULL B ("C = alpha * A^T", cuBLAS reads neither when beta is absent);
// hipBLAS rejects NULL there.  The reference's call is kept and reaches hipBLAS through this forwarder, which supplies the
// beta = 0 / B = C that cuBLAS implies.  The arithmetic (the T,T Sgemm, then an exact transposing copy) is the library's.
static hipblasStatus_t shim_sgeam(hipblasHandle_t h, hipblasOperation_t ta, hipblasOperation_t tb, int m, int n, const float *alpha,
                                  const float *A, int lda, const float *beta, const float *B, int ldb, float *C, int ldc)
{
    static const float zero = 0.f;
    return hipblasSgeam(h, ta, tb, m, n, alpha, A, lda, beta ? beta : &zero, B ? B : C, ldb, C, ldc);
}
#define hipblasSgeam shim_sgeam
#include "dense.h"
#undef hipblasSgeam

// src/data.cu:4 (not declared in data.h with this signature)
void reorderCSR(const int *ptr, const int *idx, const int *map, const int *reverse_map, int num_v, int num_e, int *&newptr, int *&newidx);

namespace {

template <class T>
T *to_dev(const T *h, size_t n)
{
    T *d = nullptr;
    if (hipMalloc((void **)&d, (n ? n : 1) * sizeof(T)) != hipSuccess) return nullptr;
    if (n && hipMemcpy(d, h, n * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return d;
}

int copy_vec(const std::vector<int> &v, int *out, int cap)
{
    if ((int)v.size() > cap) return -1;
    if (!v.empty()) memcpy(out, v.data(), v.size() * sizeof(int));
    return (int)v.size();
}

// exposes the protected schedule arrays of the reference's base class (aggregator.h:130-133)
struct GcnProbe : Aggregator_GCN {
    using Aggregator_GCN::Aggregator_GCN;
    int *sched_ptr() { return d_ptr_scheduled; }
    int *sched_target() { return d_target_scheduled; }
    int *edgelist() { return d_edgelist; }
};
struct GatProbe : Aggregator_GAT {
    using Aggregator_GAT::Aggregator_GAT;
};

}  // namespace

extern "C" {

#define REF_API __attribute__((visibility("default")))

REF_API int ref_device_count()
{
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

// ---- host-only: the reference's schedulers (graph_schedule.h).  Return the number of groups, -1 when `cap` is too small.
REF_API int ref_neighbor_grouping(const int *ptr, const int *idx, int ng, int num_v, int num_e, int *ptr_s, int *idx_s, int *target, int cap)
{
    std::vector<int> p, i, t;
    neighbor_grouping_schedule(const_cast<int *>(ptr), const_cast<int *>(idx), ng, num_v, num_e, &p, &i, &t);
    if (copy_vec(p, ptr_s, cap + 1) < 0 || copy_vec(i, idx_s, num_e) < 0 || copy_vec(t, target, cap) < 0) return -1;
    return (int)t.size();
}

REF_API int ref_locality_schedule(const int *ptr, const int *idx, int par_num, int num_v, int total_num_v, int *ptr_s, int *idx_s, int *target,
                                  int cap, int num_e)
{
    std::vector<int> p, i, t;
    locality_schedule(const_cast<int *>(ptr), const_cast<int *>(idx), par_num, num_v, &p, &i, &t, total_num_v);
    if (copy_vec(p, ptr_s, cap + 1) < 0 || copy_vec(i, idx_s, num_e) < 0 || copy_vec(t, target, cap) < 0) return -1;
    return (int)t.size();
}

REF_API int ref_locality_neighbor_grouping(const int *ptr, const int *idx, int par_num, int ng, int num_v, int total_num_v, int *ptr_s,
                                           int *idx_s, int *target, int cap, int num_e)
{
    std::vector<int> p, i, t;
    localityNeighborGrouping(const_cast<int *>(ptr), const_cast<int *>(idx), par_num, ng, num_v, &p, &i, &t, total_num_v);
    if (copy_vec(p, ptr_s, cap + 1) < 0 || copy_vec(i, idx_s, num_e) < 0 || copy_vec(t, target, cap) < 0) return -1;
    return (int)t.size();
}

// ---- host-only: src/data.cu
REF_API int ref_reorder_csr(const int *ptr, const int *idx, const int *map, const int *reverse_map, int num_v, int num_e, int *newptr, int *newidx)
{
    int *np = newptr, *ni = newidx;  // non-null: reorderCSR fills the caller's buffers
    reorderCSR(ptr, idx, map, reverse_map, num_v, num_e, np, ni);
    return 0;
}

// load_graph reads "../data/<dset>.*" relative to the working directory: `workdir` is a directory next to that data/.
// With `shuffle` and an existing "<dset>.reorder<suffix>" the graph comes back reordered and rows / reverse_rows are filled.
REF_API int ref_load_graph(const char *workdir, const char *dset, int shuffle, const char *reorder_suffix, int *num_v, int *num_e, int *ptr_out,
                           int cap_v, int *idx_out, int cap_e, int *rows_out, int *reverse_rows_out)
{
    char cwd[4096];
    if (!getcwd(cwd, sizeof(cwd)) || chdir(workdir) != 0) return -2;
    rows = nullptr;
    reverse_rows = nullptr;
    reorderfile = "";
    int v = 0, e = 0, *p = nullptr, *i = nullptr;
    load_graph(std::string(dset), v, e, p, i, shuffle != 0, std::string(reorder_suffix ? reorder_suffix : ""));
    (void)!chdir(cwd);
    *num_v = v;
    *num_e = e;
    if (v > cap_v || e > cap_e) return -1;
    memcpy(ptr_out, p, (size_t)(v + 1) * sizeof(int));
    memcpy(idx_out, i, (size_t)e * sizeof(int));
    int reordered = 0;
    if (rows && reverse_rows) {
        reordered = 1;
        if (rows_out) memcpy(rows_out, rows, (size_t)v * sizeof(int));
        if (reverse_rows_out) memcpy(reverse_rows_out, reverse_rows, (size_t)v * sizeof(int));
    }
    delete[] p;
    delete[] i;
    return reordered;
}

// ---- device: Aggregator_GCN (aggr_gcn.h:362-444).  Host arrays in / out; `scheduled` 1 runs schedule(neighbor_grouping,
// {ng}) first (aggr_gcn.h:379-410), 2 schedule(locality_neighbor_grouping, {par, ng}) (:500-537).  sched_* (optional, capacity `cap` groups) receive the reference's scheduled arrays.
// Returns num_target (>= 0), or a negative error.
REF_API int ref_gcn_run(const int *ptr, const int *idx, const float *val, int num_v, int num_e, const float *x, float *y, int feat, int block,
                        int scheduled, int ng, int *sched_ptr, int *sched_target, int cap, int par)
{
    n = num_v;
    m = num_e;
    feature_len = feat;
    int *d_ptr = to_dev(ptr, (size_t)num_v + 1), *d_idx = to_dev(idx, num_e);
    float *d_val = to_dev(val, num_e), *d_x = to_dev(x, (size_t)num_v * feat), *d_y = nullptr;
    if (!d_ptr || !d_idx || !d_val || !d_x || hipMalloc((void **)&d_y, (size_t)num_v * feat * sizeof(float)) != hipSuccess) return -2;
    if (hipMemset(d_y, 0xff, (size_t)num_v * feat * sizeof(float)) != hipSuccess) return -2;  // NaN pattern: run() must overwrite it
    int rc = 0;
    {
        GcnProbe agg(nullptr, nullptr, d_ptr, d_idx, num_v, num_e, feat, feat, d_val);  // owns d_ptr / d_idx / d_val (aggregator.h:58-66)
        if (scheduled == 2) {  // Aggregator_GCN::schedule override (aggr_gcn.h:500-537): permuted idx AND val
            int param[2] = {par, ng};
            agg.schedule(locality_neighbor_grouping, param);
        } else if (scheduled) {
            int param[2] = {ng, 0};
            agg.schedule(neighbor_grouping, param);
        }
        agg.run(d_x, d_y, block, scheduled != 0);
        if (hipDeviceSynchronize() != hipSuccess) rc = -3;
        rc = rc ? rc : agg.num_target;
        if (!rc || rc > 0) {
            if (scheduled && sched_ptr && sched_target) {
                if (agg.num_target > cap) rc = -1;
                else {
                    (void)hipMemcpy(sched_ptr, agg.sched_ptr(), (size_t)(agg.num_target + 1) * sizeof(int), hipMemcpyDeviceToHost);
                    (void)hipMemcpy(sched_target, agg.sched_target(), (size_t)agg.num_target * sizeof(int), hipMemcpyDeviceToHost);
                }
            }
            if (hipMemcpy(y, d_y, (size_t)num_v * feat * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) rc = -3;
        }
    }
    (void)hipFree(d_x);
    (void)hipFree(d_y);
    return rc;
}

// Aggregator::csr2edgelist (aggregator.h:115-122): edgelist[2e] = source, [2e + 1] = row
REF_API int ref_csr2edgelist(const int *ptr, const int *idx, int num_v, int num_e, int *edgelist)
{
    n = num_v;
    m = num_e;
    int *d_ptr = to_dev(ptr, (size_t)num_v + 1), *d_idx = to_dev(idx, num_e);
    std::vector<float> ones((size_t)num_e, 1.0f);
    float *d_val = to_dev(ones.data(), num_e);
    if (!d_ptr || !d_idx || !d_val) return -2;
    int rc = 0;
    {
        GcnProbe agg(nullptr, nullptr, d_ptr, d_idx, num_v, num_e, 32, 32, d_val);
        agg.csr2edgelist();
        if (hipDeviceSynchronize() != hipSuccess ||
            hipMemcpy(edgelist, agg.edgelist(), (size_t)2 * num_e * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess)
            rc = -3;
    }
    return rc;
}

// ---- device: Aggregator_GAT::run (aggr_gat.h:317-354): att is [V, 2] (centre term, source term); slope 0.2 (:347).
// The scheduled path never zeroes vout / scalar (aggr_gat.h:305,333): a fresh object and a zeroed vout per call here.
REF_API int ref_gat_run(const int *ptr, const int *idx, int num_v, int num_e, const float *x, const float *att, float *y, int feat, int block,
                        int scheduled, int ng)
{
    n = num_v;
    m = num_e;
    feature_len = feat;
    int *d_ptr = to_dev(ptr, (size_t)num_v + 1), *d_idx = to_dev(idx, num_e);
    float *d_x = to_dev(x, (size_t)num_v * feat), *d_att = to_dev(att, (size_t)num_v * 2), *d_y = nullptr;
    if (!d_ptr || !d_idx || !d_x || !d_att || hipMalloc((void **)&d_y, (size_t)num_v * feat * sizeof(float)) != hipSuccess) return -2;
    if (hipMemset(d_y, scheduled ? 0 : 0xff, (size_t)num_v * feat * sizeof(float)) != hipSuccess) return -2;
    int rc = 0;
    {
        GatProbe agg(nullptr, nullptr, d_ptr, d_idx, num_v, num_e, feat, feat);
        if (scheduled) {
            int param[2] = {ng, 0};
            agg.schedule(neighbor_grouping, param);
        }
        agg.run(d_x, d_att, d_y, block, scheduled != 0);
        if (hipDeviceSynchronize() != hipSuccess) rc = -3;
        rc = rc ? rc : agg.num_target;
        if (hipMemcpy(y, d_y, (size_t)num_v * feat * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) rc = -3;
    }
    (void)hipFree(d_x);
    (void)hipFree(d_att);
    (void)hipFree(d_y);
    return rc;
}

// ---- device: time the reference's run() on this GPU (hipEvents around `iters` calls after `warm` untimed ones; the
// scheduled path includes its cudaMemset of vout, aggr_gcn.h:393, as the reference's own drivers time it).  kind 0: GCN, 1: GAT
// (att = first 2 V floats of `aux`; GCN: aux = val[E]).  Returns microseconds per call in *us, num_target as the result.
REF_API int ref_time_run(int kind, const int *ptr, const int *idx, const float *aux, int num_v, int num_e, const float *x, int feat, int block,
                         int scheduled, int ng, int warm, int iters, double *us)
{
    n = num_v;
    m = num_e;
    feature_len = feat;
    int *d_ptr = to_dev(ptr, (size_t)num_v + 1), *d_idx = to_dev(idx, num_e);
    float *d_aux = to_dev(aux, kind == 0 ? (size_t)num_e : (size_t)num_v * 2), *d_x = to_dev(x, (size_t)num_v * feat), *d_y = nullptr;
    if (!d_ptr || !d_idx || !d_aux || !d_x || hipMalloc((void **)&d_y, (size_t)num_v * feat * sizeof(float)) != hipSuccess) return -2;
    (void)hipMemset(d_y, 0, (size_t)num_v * feat * sizeof(float));
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -2;
    int rc = 0;
    auto loop = [&](auto &agg, auto call) {
        if (scheduled) {
            int param[2] = {ng, 0};
            agg.schedule(neighbor_grouping, param);
        }
        for (int i = 0; i < warm; ++i) call();
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0, 0);
        for (int i = 0; i < iters; ++i) call();
        (void)hipEventRecord(e1, 0);
        if (hipEventSynchronize(e1) != hipSuccess) { rc = -3; return; }
        float ms = 0.0f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        *us = (double)ms * 1e3 / (iters > 0 ? iters : 1);
        rc = agg.num_target;
    };
    if (kind == 0) {
        GcnProbe agg(nullptr, nullptr, d_ptr, d_idx, num_v, num_e, feat, feat, d_aux);
        loop(agg, [&] { agg.run(d_x, d_y, block, scheduled != 0); });
    } else {
        GatProbe agg(nullptr, nullptr, d_ptr, d_idx, num_v, num_e, feat, feat);
        loop(agg, [&] { agg.run(d_x, d_aux, d_y, block, scheduled != 0); });
        (void)hipFree(d_aux);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(d_x);
    (void)hipFree(d_y);
    return rc;
}

// ---- device: Aggregator_GCN::runEdgeWise (aggr_gcn.h:446-460 -> aggr_gcn_edgewise :291-302; 32 columns, one 32-lane warp per
// edge, atomicAdd) and run_with_nn (:491-499 -> aggr_gcn_nn :304-359: neighbor-grouped aggregation whose group partials are
// multiplied by `weight` [feat, out] and added into `transformed` with atomics; vout / transformed zeroed here as its caller
// Figure10/main_b.cu does).  what 0: edge-wise (feat must be 32, num_e a multiple of block / 32: the kernel's bound check is
// `row > num_e`), 1: run_with_nn (out <= 32).
REF_API int ref_gcn_variant(int what, const int *ptr, const int *idx, const float *val, int num_v, int num_e, const float *x, float *y, int feat,
                            int block, int ng, const float *weight, float *transformed, int out)
{
    n = num_v;
    m = num_e;
    feature_len = feat;
    int *d_ptr = to_dev(ptr, (size_t)num_v + 1), *d_idx = to_dev(idx, num_e);
    float *d_val = to_dev(val, num_e), *d_x = to_dev(x, (size_t)num_v * feat), *d_y = nullptr, *d_w = nullptr, *d_t = nullptr;
    if (!d_ptr || !d_idx || !d_val || !d_x || hipMalloc((void **)&d_y, (size_t)num_v * feat * sizeof(float)) != hipSuccess) return -2;
    (void)hipMemset(d_y, 0, (size_t)num_v * feat * sizeof(float));
    if (what == 1) {
        d_w = to_dev(weight, (size_t)feat * out);
        if (!d_w || hipMalloc((void **)&d_t, (size_t)num_v * out * sizeof(float)) != hipSuccess) return -2;
        (void)hipMemset(d_t, 0, (size_t)num_v * out * sizeof(float));
    }
    int rc = 0;
    {
        GcnProbe agg(nullptr, nullptr, d_ptr, d_idx, num_v, num_e, feat, what == 1 ? out : feat, d_val);
        if (what == 0) {
            agg.runEdgeWise(d_x, d_y, block, false);
        } else {
            int param[2] = {ng, 0};
            agg.schedule(neighbor_grouping, param);
            agg.run_with_nn(d_x, d_y, d_w, d_t, block);
        }
        if (hipDeviceSynchronize() != hipSuccess) rc = -3;
        if (hipMemcpy(y, d_y, (size_t)num_v * feat * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) rc = -3;
        if (what == 1 && hipMemcpy(transformed, d_t, (size_t)num_v * out * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) rc = -3;
    }
    (void)hipFree(d_x); (void)hipFree(d_y); (void)hipFree(d_w); (void)hipFree(d_t);
    return rc;
}

// ---- device: matmul_NN (dense.h:4-23): C[M, N] = A[M, K] . B[K, N], all row-major, through the vendor GEMM (T,T into a column-major
// scratch) and a transposing geam.  The handle is made the way Figure10/main_b.cu:31 makes it.
REF_API int ref_matmul_nn(const float *a, const float *b, float *c, int M, int N, int K)
{
    float *d_a = to_dev(a, (size_t)M * K), *d_b = to_dev(b, (size_t)K * N), *d_c = nullptr, *d_t = nullptr;
    if (!d_a || !d_b || hipMalloc((void **)&d_c, (size_t)M * N * sizeof(float)) != hipSuccess ||
        hipMalloc((void **)&d_t, (size_t)M * N * sizeof(float)) != hipSuccess) return -2;
    static bool have_handle = false;   // util.cu:175 new[]s the handle array without initialising it
    if (!have_handle && hipblasCreate(&cublasHs[0]) != HIPBLAS_STATUS_SUCCESS) return -4;
    have_handle = true;
    matmul_NN(d_a, d_b, d_c, M, N, K, d_t);
    int rc = 0;
    if (hipDeviceSynchronize() != hipSuccess) rc = -3;
    if (hipMemcpy(c, d_c, (size_t)M * N * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) rc = -3;
    (void)hipFree(d_a); (void)hipFree(d_b); (void)hipFree(d_c); (void)hipFree(d_t);
    return rc;
}

// ---- device: the naive SpMM and the validators (spmm.h:11-91,223-265).  spmm<L>: one thread per row, y rows of empty rows
// are left untouched (the caller's y comes back for them).  L in {32, 64, 128}.
REF_API int ref_spmm_naive(const int *ptr, const int *idx, const float *val, int num_v, int num_e, const float *x, float *y, int feat)
{
    int *d_ptr = to_dev(ptr, (size_t)num_v + 1), *d_idx = to_dev(idx, num_e);
    float *d_val = to_dev(val, num_e), *d_x = to_dev(x, (size_t)num_v * feat), *d_y = to_dev(y, (size_t)num_v * feat);
    if (!d_ptr || !d_idx || !d_val || !d_x || !d_y) return -2;
    const int grid = (num_v + TB - 1) / TB;
    switch (feat) {
        case 32: spmm<32><<<grid, TB>>>(num_v, d_ptr, d_idx, d_val, d_x, d_y); break;
        case 64: spmm<64><<<grid, TB>>>(num_v, d_ptr, d_idx, d_val, d_x, d_y); break;
        case 128: spmm<128><<<grid, TB>>>(num_v, d_ptr, d_idx, d_val, d_x, d_y); break;
        default: return -1;
    }
    int rc = hipDeviceSynchronize() == hipSuccess && hipMemcpy(y, d_y, (size_t)num_v * feat * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -3;
    (void)hipFree(d_ptr); (void)hipFree(d_idx); (void)hipFree(d_val); (void)hipFree(d_x); (void)hipFree(d_y);
    return rc;
}

// valid (spmm.h:35-71) / validReordered (:73-91; `map` = the global rows[] of the loader, NULL: plain valid)
REF_API int ref_valid(const float *ref_y, const float *ans, const int *map, int num_v, int feat)
{
    float *d_r = to_dev(ref_y, (size_t)num_v * feat), *d_a = to_dev(ans, (size_t)num_v * feat);
    if (!d_r || !d_a) return -2;
    int *saved = rows;
    rows = const_cast<int *>(map);
    const int bad = map ? validReordered(d_r, d_a, num_v, feat) : valid(d_r, d_a, num_v * feat);
    rows = saved;
    (void)hipFree(d_r); (void)hipFree(d_a);
    return bad;
}

// ---- device: the edge-softmax stages (aggr_gat.h:395-425), each through the reference's own method with BLOCK_SIZE = 32
// (see the header).  what: 0 run_att (att[V,2] -> val[E] normalised), 1 run_u_add_v (att[V,2] -> val[E]),
// 2 run_add_to_center (val[E] -> vec[V] row sums), 3 run_div_each (vec[V], val[E] in/out).
REF_API int ref_gat_edge_stage(int what, const int *ptr, const int *idx, int num_v, int num_e, float *att_or_vec, float *val)
{
    n = num_v;
    m = num_e;
    int *d_ptr = to_dev(ptr, (size_t)num_v + 1), *d_idx = to_dev(idx, num_e);
    const size_t nv = (what == 0 || what == 1) ? (size_t)num_v * 2 : (size_t)num_v;
    float *d_v = to_dev(att_or_vec, nv), *d_e = to_dev(val, num_e);
    if (!d_ptr || !d_idx || !d_v || !d_e) return -2;
    int rc = 0;
    {
        GatProbe agg(nullptr, nullptr, d_ptr, d_idx, num_v, num_e, 32, 32);
        switch (what) {
            case 0: agg.run_att(d_v, d_e, 32); break;
            case 1: agg.run_u_add_v(d_v, d_e, 32); break;
            case 2: agg.run_add_to_center(d_e, d_v, 32); break;
            default: agg.run_div_each(d_v, d_e, 32); break;
        }
        if (hipDeviceSynchronize() != hipSuccess) rc = -3;
        if (what == 2) { if (hipMemcpy(att_or_vec, d_v, (size_t)num_v * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) rc = -3; }
        else if (hipMemcpy(val, d_e, (size_t)num_e * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) rc = -3;
    }
    (void)hipFree(d_v);
    (void)hipFree(d_e);
    return rc;
}

}  // extern "C"
