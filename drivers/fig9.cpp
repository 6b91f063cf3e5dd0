// fig9 -- GCN aggregation micro-benchmark, same flags and call sequence as the reference's
// Figure9/main.cu:15-75:  fig9.out --dataset D --feature-len F [--datadir DIR] [--reorder _thres_0.2] [--nei NG]
// 10 un-scheduled runs, 10 neighbor-grouped runs (+ the library's balanced mode), timed per iteration with
// hipEvents; results as JSON lines on stderr.
#include "../include/compat/aggr_gcn.h"
#include "../include/compat/sample.h"
#include "../include/compat/spmm.h"
#include "common.h"

int main(int argc, char **argv)
{
    const int times = 10;
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
    const int BLOCK_SIZE = 512;
    dbg(NEIGHBOR_NUM);

    auto g = fullGraph(gptrs[0], gidxs[0]);
    Aggregator_GCN *atgcn = new Aggregator_GCN(g, feature_len, feature_len, val);
    int tmparr[] = {NEIGHBOR_NUM};
    timestamp(ts0);
    atgcn->schedule(neighbor_grouping, tmparr);
    timestamp(ts1);
    double neighbor_grouping_schedule_time = getDuration(ts0, ts1);
    dbg(neighbor_grouping_schedule_time);

    for (int i = 0; i < times; ++i) atgcn->run(x, y, BLOCK_SIZE, 0);  // warm-up
    checkHipErrors(hipDeviceSynchronize());
    report("unscheduled (CSR rows)", median_time(times, [&] { atgcn->run(x, y, BLOCK_SIZE, 0); }));
    report("neighbor_grouping", median_time(times, [&] { atgcn->run(x, y2, BLOCK_SIZE, 1); }));
    const int mismatches = valid(y, y2, n * feature_len);  // reference validator: rel err > 1e-2
    dbg(mismatches);
    atgcn->run_balanced(x, y2, feature_len);
    report("balanced", median_time(times, [&] { atgcn->run_balanced(x, y2, feature_len); }));
    delete atgcn;
    safeFree(x);
    safeFree(y);
    safeFree(y2);
    // near-zero sums of N(0,1) products can exceed the validator's 1e-2 *relative* bound by rounding alone;
    // more than 1e-4 of the elements would mean a real defect
    return mismatches <= (long)n * feature_len / 10000 ? 0 : 2;
}
