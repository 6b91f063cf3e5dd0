// dist_step -- the row-partitioned (multi-GPU) aggregation step driven from C++ through the C-ABI only: one process per
// GPU, RCCL grouped send/recv behind gnnagg_dist_* (the reference itself asserts GPUNUM == 1,
// Figure9/main.cu:19).  Start it once per GPU with RANK / WORLD_SIZE / LOCAL_RANK in the environment (the variables
// torchrun and most MPI launchers set), e.g.
//     for r in 0 1 2 3 4 5 6 7; do RANK=$r WORLD_SIZE=8 LOCAL_RANK=$r ./dist_step.out --dataset products --datadir D \
//         --feature-len 100 --idfile /tmp/gnnagg.id & done; wait
// Every rank loads the graph, keeps ITS row slice (gnnagg_partition_rows + gnnagg_halo_plan_slice), exchanges the request
// lists once (gnnagg_dist_alltoallv), then times `--iters` steps.  --plan overlap (default): ONE host call per step,
// gnnagg_dist_step_gcn -- pack + grouped send / recv on the step's communication stream, the local-source edges beside it, the
// halo-source edges after the join (GNNAGG_FLAG_ACCUMULATE).  --plan onepass: halo pull (gnnagg_dist_halo_exchange), then the
// unchanged single-GPU kernel over [X_local ; X_halo].  One JSON line per rank on stderr; rank 0 also prints the slowest rank's time.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../include/gnnagg.h"

#define CK(expr)                                                                                   \
    do {                                                                                           \
        const int rc_ = (expr);                                                                    \
        if (rc_ != GNNAGG_OK) {                                                                    \
            fprintf(stderr, "%s failed (%d): %s\n", #expr, rc_, gnnagg_last_error());              \
            exit(1);                                                                               \
        }                                                                                          \
    } while (0)
#define HCK(expr)                                                                                  \
    do {                                                                                           \
        const hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "%s failed: %s\n", #expr, hipGetErrorString(e_));                      \
            exit(1);                                                                               \
        }                                                                                          \
    } while (0)

static int env_int(const char *name, int dflt) { const char *v = getenv(name); return v ? atoi(v) : dflt; }

template <class T>
static T *to_device(const std::vector<T> &h)
{
    T *p = nullptr;
    HCK(hipMalloc((void **)&p, sizeof(T) * std::max<size_t>(h.size(), 1)));
    if (!h.empty()) HCK(hipMemcpy(p, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice));
    return p;
}

// --check: feature c of global row g (exactly representable, so the halo comparison is bit-exact)
static float closed_form(int g, int c) { return (float)(((long)g * 131 + (long)c * 71) % 1013) / 1013.0f - 0.5f; }

int main(int argc, char **argv)
{
    std::string dataset, datadir = "../data/", idfile = "/tmp/gnnagg_dist.id", plan = "overlap";
    std::string stages = "1";   // staged exchange (overlap plan): K stripes of every peer's rows, or "owner" (one ring distance per stage)
    int feat = 128, iters = 20, check = 0;   // --check 1: features are a closed form of the GLOBAL row id; halo rows and results verified on the host
    for (int i = 1; i + 1 < argc; i += 2) {
        const std::string k = argv[i];
        if (k == "--dataset") dataset = argv[i + 1];
        else if (k == "--datadir") datadir = argv[i + 1];
        else if (k == "--feature-len") feat = atoi(argv[i + 1]);
        else if (k == "--iters") iters = atoi(argv[i + 1]);
        else if (k == "--idfile") idfile = argv[i + 1];
        else if (k == "--plan") plan = argv[i + 1];
        else if (k == "--stages") stages = argv[i + 1];
        else if (k == "--check") check = atoi(argv[i + 1]);
        else { fprintf(stderr, "unknown flag %s\n", k.c_str()); return 2; }
    }
    if (dataset.empty()) { fprintf(stderr, "usage: dist_step.out --dataset D [--datadir DIR] [--feature-len F] [--iters K] [--idfile PATH] [--plan overlap|onepass] [--stages K|owner] [--check 1]\n"); return 2; }
    const int rank = env_int("RANK", 0), world = env_int("WORLD_SIZE", 1), local_rank = env_int("LOCAL_RANK", 0);
    HCK(hipSetDevice(local_rank));
    gnnagg_set_abort_on_error(0);

    // the graph and this rank's slice of it
    int V = 0, E = 0, *h_ptr = nullptr, *h_idx = nullptr;
    CK(gnnagg_load_graph(datadir.c_str(), dataset.c_str(), "", 0, &V, &E, &h_ptr, &h_idx, nullptr, nullptr));
    std::vector<int> bounds((size_t)world + 1);
    CK(gnnagg_partition_rows(h_ptr, V, world, bounds.data()));
    const int r0 = bounds[rank], r1 = bounds[rank + 1], n_local = r1 - r0;
    const int nnz = h_ptr[r1] - h_ptr[r0];
    std::vector<int> lptr((size_t)n_local + 1), lidx((size_t)std::max(nnz, 1)), recv_rows_i((size_t)world);
    int *halo_ids = nullptr, n_halo = 0;
    CK(gnnagg_halo_plan_slice(h_ptr + r0, h_idx + h_ptr[r0], V, bounds.data(), world, rank, lptr.data(), lidx.data(), &halo_ids,
                              recv_rows_i.data(), &n_halo));

    gnnagg_comm comm = 0;
    CK(gnnagg_dist_comm_create_from_file(idfile.c_str(), rank, world, 300, &comm));
    hipStream_t stream;
    HCK(hipStreamCreate(&stream));

    // one-time: tell every owner which of its rows this rank needs (counts, then the id lists)
    std::vector<long long> recv_rows(world), send_rows(world), ones(world, 1);
    for (int p = 0; p < world; ++p) recv_rows[p] = recv_rows_i[p];
    {
        long long *d_a = to_device(recv_rows), *d_b = to_device(send_rows);
        CK(gnnagg_dist_alltoallv(comm, d_a, ones.data(), d_b, ones.data(), (int)sizeof(long long), stream));
        HCK(hipStreamSynchronize(stream));
        HCK(hipMemcpy(send_rows.data(), d_b, sizeof(long long) * world, hipMemcpyDeviceToHost));
        (void)hipFree(d_a); (void)hipFree(d_b);
    }
    long long n_send = 0;
    for (int p = 0; p < world; ++p) n_send += send_rows[p];
    std::vector<int> h_req(halo_ids, halo_ids + n_halo);
    int *d_req = to_device(h_req), *d_send_ids = nullptr;
    HCK(hipMalloc((void **)&d_send_ids, sizeof(int) * std::max<long long>(n_send, 1)));
    CK(gnnagg_dist_alltoallv(comm, d_req, recv_rows.data(), d_send_ids, send_rows.data(), (int)sizeof(int), stream));
    HCK(hipStreamSynchronize(stream));
    // staged exchange (gnnagg_halo_stage_plan): the halo tail and the send buffer become stage-major.  Requests went out owner-major
    // in ascending id = the order the rows arrive in (a stage takes a contiguous slice of every list), so only the slots move.
    const int stage_mode = stages == "owner" ? GNNAGG_STAGES_OWNER : GNNAGG_STAGES_STRIPE;
    const int stage_k = stages == "owner" ? 1 : std::max(1, atoi(stages.c_str()));
    int n_stages = 1;
    CK(gnnagg_halo_stage_plan(nullptr, nullptr, world, rank, stage_mode, stage_k, &n_stages, nullptr, nullptr, nullptr, nullptr));
    if (plan != "overlap") n_stages = 1;
    std::vector<long long> stage_recv((size_t)n_stages * world), stage_send((size_t)n_stages * world);
    std::vector<int> new_of_old((size_t)std::max(n_halo, 1)), send_order((size_t)std::max<long long>(n_send, 1));
    if (n_stages > 1) {
        CK(gnnagg_halo_stage_plan(recv_rows.data(), send_rows.data(), world, rank, stage_mode, stage_k, &n_stages, stage_recv.data(), new_of_old.data(),
                                  stage_send.data(), send_order.data()));
        for (auto &c : lidx)
            if (c >= n_local) c = n_local + new_of_old[(size_t)(c - n_local)];
    } else {
        for (int p = 0; p < world; ++p) { stage_recv[p] = recv_rows[p]; stage_send[p] = send_rows[p]; }
    }
    {   // global ids -> rows of the local X, in the order the send buffer is packed in (stage-major)
        std::vector<int> ids((size_t)n_send), staged((size_t)n_send);
        HCK(hipMemcpy(ids.data(), d_send_ids, sizeof(int) * n_send, hipMemcpyDeviceToHost));
        for (long long i = 0; i < n_send; ++i) staged[(size_t)i] = ids[(size_t)(n_stages > 1 ? send_order[(size_t)i] : i)] - r0;
        HCK(hipMemcpy(d_send_ids, staged.data(), sizeof(int) * n_send, hipMemcpyHostToDevice));
    }

    // device state: local CSR (columns = X_ext slots), X_ext = [X_local ; X_halo], send buffer, Y
    int *d_ptr = to_device(lptr), *d_idx = to_device(lidx);
    std::vector<float> hx((size_t)n_local * feat);
    {
        std::mt19937_64 gen(123 + rank);
        std::normal_distribution<float> nd(0.f, 1.f);
        for (auto &v : hx) v = nd(gen);
        if (check)
            for (int r = 0; r < n_local; ++r)
                for (int c = 0; c < feat; ++c) hx[(size_t)r * feat + c] = closed_form(r0 + r, c);
    }
    float *d_x_ext = nullptr, *d_send = nullptr, *d_y = nullptr;
    HCK(hipMalloc((void **)&d_x_ext, sizeof(float) * std::max<size_t>((size_t)(n_local + n_halo) * feat, 1)));
    HCK(hipMalloc((void **)&d_send, sizeof(float) * std::max<size_t>((size_t)n_send * feat, 1)));
    HCK(hipMalloc((void **)&d_y, sizeof(float) * std::max<size_t>((size_t)n_local * feat, 1)));
    HCK(hipMemcpy(d_x_ext, hx.data(), sizeof(float) * hx.size(), hipMemcpyHostToDevice));
    float *d_x_halo = d_x_ext + (size_t)n_local * feat;

    gnnagg_handle agg = 0, agg_loc = 0;
    std::vector<gnnagg_handle> agg_rem((size_t)n_stages, 0);
    gnnagg_dist_step_t dstep = 0;
    int *d_pl = nullptr, *d_il = nullptr;
    std::vector<int *> d_stage_arrays;
    if (plan == "overlap") {
        // the rank's CSR as 1 + S: edges whose source is an owned row (columns = local rows) and, per stage, the edges whose source
        // is a halo row that arrives in that stage (columns = halo slots); in-row order kept
        std::vector<long long> stage0((size_t)n_stages + 1, 0);
        for (int st = 0; st < n_stages; ++st) {
            stage0[(size_t)st + 1] = stage0[(size_t)st];
            for (int p = 0; p < world; ++p) stage0[(size_t)st + 1] += stage_recv[(size_t)st * world + p];
        }
        auto stage_of = [&](int slot) { int st = 0; while (st + 1 < n_stages && slot >= stage0[(size_t)st + 1]) ++st; return st; };
        std::vector<int> pl((size_t)n_local + 1, 0), il;
        std::vector<std::vector<int>> pr((size_t)n_stages, std::vector<int>((size_t)n_local + 1, 0)), ir((size_t)n_stages);
        for (int r = 0; r < n_local; ++r) {
            for (int e = lptr[r]; e < lptr[r + 1]; ++e) {
                if (lidx[e] < n_local) il.push_back(lidx[e]);
                else ir[(size_t)stage_of(lidx[e] - n_local)].push_back(lidx[e] - n_local);
            }
            pl[r + 1] = (int)il.size();
            for (int st = 0; st < n_stages; ++st) pr[(size_t)st][r + 1] = (int)ir[(size_t)st].size();
        }
        d_pl = to_device(pl); d_il = to_device(il);
        CK(gnnagg_gcn_create(d_pl, d_il, nullptr, n_local, (int)il.size(), &agg_loc));
        CK(gnnagg_schedule_balanced(agg_loc, 0));
        for (int st = 0; st < n_stages; ++st) {
            if (ir[(size_t)st].empty()) continue;
            int *dp = to_device(pr[(size_t)st]), *di = to_device(ir[(size_t)st]);
            d_stage_arrays.push_back(dp); d_stage_arrays.push_back(di);
            CK(gnnagg_gcn_create(dp, di, nullptr, n_local, (int)ir[(size_t)st].size(), &agg_rem[(size_t)st]));
            CK(gnnagg_schedule_balanced(agg_rem[(size_t)st], 0));
        }
        CK(gnnagg_dist_step_create_staged(comm, agg_loc, n_stages, agg_rem.data(), d_send_ids, stage_send.data(), stage_recv.data(), &dstep));
    } else {
        CK(gnnagg_gcn_create(d_ptr, d_idx, nullptr, n_local, nnz, &agg));
        CK(gnnagg_set_stream(agg, stream));
        CK(gnnagg_schedule_balanced(agg, 0));
    }
    auto step = [&] {
        if (dstep) {
            CK(gnnagg_dist_step_gcn(dstep, d_x_ext, d_x_halo, d_send, d_y, feat, GNNAGG_REDUCE_SUM, stream));
        } else {
            CK(gnnagg_dist_halo_exchange(comm, d_x_ext, d_send_ids, send_rows.data(), recv_rows.data(), feat, d_send, d_x_halo, stream));
            CK(gnnagg_gcn_run(agg, d_x_ext, d_y, feat, GNNAGG_MODE_BALANCED, GNNAGG_REDUCE_SUM));
        }
    };
    for (int i = 0; i < 3; ++i) step();
    HCK(hipStreamSynchronize(stream));
    if (check) {
        // every rank checks, without any further exchange: the halo rows it pulled against the closed form of their global ids,
        // and its result rows against the sum over its CSR slice computed on the host (unit weights, double accumulation)
        std::vector<int> slot_gid((size_t)std::max(n_halo, 1));
        for (int i = 0; i < n_halo; ++i) slot_gid[(size_t)(n_stages > 1 ? new_of_old[(size_t)i] : i)] = halo_ids[i];
        std::vector<float> got_halo((size_t)n_halo * feat), got_y((size_t)n_local * feat);
        HCK(hipMemcpy(got_halo.data(), d_x_halo, sizeof(float) * got_halo.size(), hipMemcpyDeviceToHost));
        HCK(hipMemcpy(got_y.data(), d_y, sizeof(float) * got_y.size(), hipMemcpyDeviceToHost));
        long bad_halo = 0, bad_y = 0;
        for (int sl = 0; sl < n_halo; ++sl)
            for (int c = 0; c < feat; ++c) bad_halo += got_halo[(size_t)sl * feat + c] != closed_form(slot_gid[(size_t)sl], c);
        for (int r = 0; r < n_local; ++r)
            for (int c = 0; c < feat; c += 7) {
                double ref = 0.0, mag = 0.0;
                for (int e = lptr[r]; e < lptr[r + 1]; ++e) {
                    const int gid = lidx[e] < n_local ? r0 + lidx[e] : slot_gid[(size_t)(lidx[e] - n_local)];
                    ref += closed_form(gid, c);
                    mag += std::fabs(closed_form(gid, c));
                }
                bad_y += std::fabs((double)got_y[(size_t)r * feat + c] - ref) > 1e-5 * mag + 1e-30;
            }
        fprintf(stderr, "{\"check\": \"rank %d\", \"halo_rows\": %d, \"bad_halo_values\": %ld, \"bad_results\": %ld}\n", rank, n_halo, bad_halo, bad_y);
        if (bad_halo || bad_y) return 3;
    }
    hipEvent_t a, b;
    HCK(hipEventCreate(&a)); HCK(hipEventCreate(&b));
    HCK(hipEventRecord(a, stream));
    for (int i = 0; i < iters; ++i) step();
    HCK(hipEventRecord(b, stream));
    HCK(hipEventSynchronize(b));
    float ms = 0.f;
    HCK(hipEventElapsedTime(&ms, a, b));
    const double sec = ms * 1e-3 / iters;
    // a checksum of the halo rows against what their owners hold is the launcher's job at world > 1; at world == 1 there is
    // no halo and the step is the single-GPU aggregation
    fprintf(stderr, "{\"plan\": \"%s\", \"stages\": %d, \"rank\": %d, \"world\": %d, \"n_local\": %d, \"nnz_local\": %d, \"n_halo\": %d, \"n_send\": %lld, \"seconds\": %.9f, "
                    "\"edges_per_s\": %.6e, \"halo_bytes\": %.0f}\n",
            plan.c_str(), n_stages, rank, world, n_local, nnz, n_halo, n_send, sec, (double)nnz / sec, (double)n_halo * feat * 4.0);
    // the slowest rank bounds the step: gather the times on rank 0
    {
        std::vector<double> mine(1, sec), all((size_t)world, 0.0);
        std::vector<long long> sc(world, 0), rc(world, 0);
        sc[0] = 1;                       // everybody sends one double to rank 0
        if (rank == 0) for (int p = 0; p < world; ++p) rc[p] = 1;
        double *d_m = to_device(mine), *d_all = to_device(all);
        CK(gnnagg_dist_alltoallv(comm, d_m, sc.data(), d_all, rc.data(), (int)sizeof(double), stream));
        HCK(hipStreamSynchronize(stream));
        if (rank == 0) {
            HCK(hipMemcpy(all.data(), d_all, sizeof(double) * world, hipMemcpyDeviceToHost));
            const double worst = *std::max_element(all.begin(), all.end());
            fprintf(stderr, "{\"summary\": \"slowest rank\", \"world\": %d, \"seconds\": %.9f, \"edges_per_s\": %.6e}\n", world, worst, (double)E / worst);
        }
        (void)hipFree(d_m); (void)hipFree(d_all);
    }
    if (dstep) CK(gnnagg_dist_step_destroy(dstep));
    if (agg) CK(gnnagg_destroy(agg));
    if (agg_loc) CK(gnnagg_destroy(agg_loc));
    for (gnnagg_handle hr : agg_rem)
        if (hr) CK(gnnagg_destroy(hr));
    CK(gnnagg_dist_comm_destroy(comm));
    gnnagg_free_host(halo_ids); gnnagg_free_host(h_ptr); gnnagg_free_host(h_idx);
    if (rank == 0) remove(idfile.c_str());
    return 0;
}
