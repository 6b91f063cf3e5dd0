// fig10a -- GAT: 3-kernel base vs adapter (run_att + gcn.run) vs fused gat.run, same flags and call sequence
// as the reference's Figure10/main_a.cu:19-114:  fig10a.out --dataset D --feature-len F [--nei NG] [--datadir DIR]
#include "../include/compat/aggr_gat.h"
#include "../include/compat/aggr_gcn.h"
#include "../include/compat/sample.h"
#include "common.h"

__global__ void exp_leaky(float *v, int count, float slope)
{
    // the exp(leaky_relu) the reference's base variant leaves to PyTorch between its kernels (our.py:145-151)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) {
        const float s = v[i];
        v[i] = expf(s > s * slope ? s : s * slope);
    }
}

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
    registerPtr(gptrs[0]);  // two aggregators share the CSR (main_a.cu:66-70)
    registerPtr(gidxs[0]);

    float *x = device_normal((size_t)n * feature_len, 123);
    float *y = device_normal((size_t)n * feature_len, 124);
    float *y2 = device_normal((size_t)n * feature_len, 125);
    float *att = device_normal((size_t)n * 2, 126);
    float *out_att = device_normal((size_t)n * 2, 127);
    float *val = device_normal((size_t)m, 128);
    registerPtr(val);

    int NEIGHBOR_NUM = 16;
    if (NEINUM != -1) NEIGHBOR_NUM = NEINUM;
    const int BLOCK_SIZE = 128;
    auto g = fullGraph(gptrs[0], gidxs[0]);
    Aggregator_GCN *atgcn = new Aggregator_GCN(g, feature_len, feature_len, val);
    Aggregator_GAT *atgat = new Aggregator_GAT(g, feature_len, feature_len);
    int tmparr[] = {NEIGHBOR_NUM};
    atgcn->schedule(neighbor_grouping, tmparr);
    atgat->schedule(neighbor_grouping, tmparr);

    for (int i = 0; i < times; ++i) atgcn->run(x, y, BLOCK_SIZE, 1);  // warm-up
    checkHipErrors(hipDeviceSynchronize());

    report("base (u_add_v, exp, add_to_center, div_each, gcn.run)", median_time(times, [&] {
               atgat->run_u_add_v(att, val, BLOCK_SIZE);
               hipLaunchKernelGGL(exp_leaky, dim3((m + 255) / 256), dim3(256), 0, nullptr, val, m, 0.2f);
               atgat->run_add_to_center(val, out_att, BLOCK_SIZE);
               atgat->run_div_each(out_att, val, BLOCK_SIZE);
               atgcn->updateval(val);
               atgcn->run(x, y, BLOCK_SIZE, 1);
           }));
    report("adapter (run_att + gcn.run)", median_time(times, [&] {
               atgat->run_att(att, val, BLOCK_SIZE);
               atgcn->updateval(val);
               atgcn->run(x, y2, BLOCK_SIZE, 1);
           }));
    report("fused (gat.run scheduled)", median_time(times, [&] { atgat->run(x, att, y2, BLOCK_SIZE, 1); }));
    report("fused (gat.run balanced)", median_time(times, [&] { atgat->run_heads(x, att, y2, feature_len, 1); }));
    return 0;
}
