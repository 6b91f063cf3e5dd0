// fig10b -- GCN aggregation + dense combine, same flags and call sequence as the reference's
// Figure10/main_b.cu:21-104:  fig10b.out --dataset D --feature-len F --outfea OUT [--nei NG] [--datadir DIR]
// base = run(scheduled) + matmul_NN; "linear fusion" entry point = run_with_nn.
#include "../include/compat/aggr_gcn.h"
#include "../include/compat/dense.h"
#include "../include/compat/sample.h"
#include "common.h"

int main(int argc, char **argv)
{
    const int times = 10;
    strip_dump_flag(argc, argv);
    argParse(argc, argv);
    const int out_feature_len = outfea;
    assert(out_feature_len > 0);
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
    float *weight = device_normal((size_t)feature_len * out_feature_len, 127);
    float *transformed = device_normal((size_t)n * out_feature_len, 128);
    float *transformed2 = device_normal((size_t)n * out_feature_len, 129);

    int NEIGHBOR_NUM = 16;
    if (NEINUM != -1) NEIGHBOR_NUM = NEINUM;
    const int BLOCK_SIZE = 128;
    auto g = fullGraph(gptrs[0], gidxs[0]);
    Aggregator_GCN *atgcn = new Aggregator_GCN(g, feature_len, out_feature_len, val);
    int tmparr[] = {NEIGHBOR_NUM};
    atgcn->schedule(neighbor_grouping, tmparr);
    for (int i = 0; i < times; ++i) atgcn->run(x, y, BLOCK_SIZE, 1);  // warm-up
    checkHipErrors(hipDeviceSynchronize());

    report("base (run + matmul_NN)", median_time(times, [&] {
               atgcn->run(x, y2, BLOCK_SIZE, 1);
               matmul_NN(y2, weight, transformed2, n, out_feature_len, feature_len, nullptr);
           }));
    report("run_with_nn", median_time(times, [&] { atgcn->run_with_nn(x, y, weight, transformed, BLOCK_SIZE); }));
    report("matmul_NN alone", median_time(times, [&] { matmul_NN(y2, weight, transformed2, n, out_feature_len, feature_len, nullptr); }));
    return 0;
}
