// fig8 -- load-balance study, same flags and call sequence as the reference's Figure8/main.cu:27-202:
//   fig8.out --dataset D --feature-len F [--nei NG] [--datadir DIR]
// run_clock on the neighbor-grouped and on the un-scheduled kernel; per-workgroup (start, end, CU) stamps are turned into
// "actual" time (first start -> last end) and "balanced" time (sum of workgroup durations / resident workgroups of the
// whole chip), the two curves of the paper's Figure 8.
#include "../include/compat/aggr_gcn.h"
#include "../include/compat/sample.h"
#include "common.h"

static void analyse(const char *what, const std::vector<clocktype> &t, int nb, double wall_s)
{
    const double hz = (double)gnnagg_wall_clock_hz();
    clocktype first = ~0ULL, last = 0;
    double sum = 0;
    std::vector<int> cus;
    for (int b = 0; b < nb; ++b) {
        if (t[3 * b + 1] == 0) continue;  // workgroup without work
        first = std::min(first, t[3 * b]);
        last = std::max(last, t[3 * b + 1]);
        sum += (double)(t[3 * b + 1] - t[3 * b]);
        cus.push_back((int)t[3 * b + 2]);
    }
    std::sort(cus.begin(), cus.end());
    const int n_cus = (int)(std::unique(cus.begin(), cus.end()) - cus.begin());
    hipDeviceProp_t prop;
    checkHipErrors(hipGetDeviceProperties(&prop, 0));
    const int resident = prop.multiProcessorCount * 6;  // 256-thread workgroups resident per CU at the kernel's register budget
    fprintf(stderr, "{\"variant\": \"%s\", \"workgroups\": %d, \"cus_seen\": %d, \"host_seconds\": %.9f, \"actual_seconds\": %.9f, "
                    "\"balanced_seconds\": %.9f}\n",
            what, nb, n_cus, wall_s, (double)(last - first) / hz, sum / hz / resident);
}

int main(int argc, char **argv)
{
    strip_dump_flag(argc, argv);
    argParse(argc, argv);
    assert(GPUNUM == 1);
    int *tmp1 = nullptr, *tmp2 = nullptr;
    load_graph(inputgraph, n, m, tmp1, tmp2);
    gptrs = new int *[1];
    gidxs = new int *[1];
    checkHipErrors(hipMalloc2((void **)gptrs, (n + 1) * sizeof(int)));
    checkHipErrors(hipMalloc2((void **)gidxs, (m > 0 ? m : 1) * sizeof(int)));
    checkHipErrors(hipMemcpy(gptrs[0], tmp1, sizeof(int) * (n + 1), hipMemcpyHostToDevice));
    checkHipErrors(hipMemcpy(gidxs[0], tmp2, sizeof(int) * m, hipMemcpyHostToDevice));
    float *x = device_normal((size_t)n * feature_len, 123);
    float *y = device_normal((size_t)n * feature_len, 124);
    float *y2 = device_normal((size_t)n * feature_len, 125);
    float *val = device_normal((size_t)m, 126);

    int NEIGHBOR_NUM = 16;
    if (NEINUM != -1) NEIGHBOR_NUM = NEINUM;
    const int BLOCK_SIZE = 64;
    auto g = fullGraph(gptrs[0], gidxs[0]);
    Aggregator_GCN *atgcn = new Aggregator_GCN(g, feature_len, feature_len, val);
    int tmparr[] = {NEIGHBOR_NUM};
    atgcn->schedule(neighbor_grouping, tmparr);

    const int nb_fine = atgcn->clock_blocks(true), nb_coarse = atgcn->clock_blocks(false);
    clocktype *thetimer = nullptr, *thetimer2 = nullptr;
    checkHipErrors(hipMalloc2((void **)&thetimer, sizeof(clocktype) * 3 * std::max(nb_fine, 1)));
    checkHipErrors(hipMalloc2((void **)&thetimer2, sizeof(clocktype) * 3 * std::max(nb_coarse, 1)));
    atgcn->run(x, y, BLOCK_SIZE, 1);  // warm-up
    const double t_fine = atgcn->run_clock(x, y, thetimer, BLOCK_SIZE, 1);
    const double t_coarse = atgcn->run_clock(x, y2, thetimer2, BLOCK_SIZE, 0);
    std::vector<clocktype> h1(3 * (size_t)std::max(nb_fine, 1)), h2(3 * (size_t)std::max(nb_coarse, 1));
    checkHipErrors(hipMemcpy(h1.data(), thetimer, sizeof(clocktype) * 3 * nb_fine, hipMemcpyDeviceToHost));
    checkHipErrors(hipMemcpy(h2.data(), thetimer2, sizeof(clocktype) * 3 * nb_coarse, hipMemcpyDeviceToHost));
    analyse("base (un-scheduled rows)", h2, nb_coarse, t_coarse);
    analyse("NG (neighbor grouping)", h1, nb_fine, t_fine);
    return 0;
}
