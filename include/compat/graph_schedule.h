// graph_schedule.h -- enum Schedule and the three host schedulers with the reference's signatures
// (reference include/graph_schedule.h:8-14, :17, :91, :156), forwarding to the C-ABI host functions.
#ifndef GNNAGG_COMPAT_GRAPH_SCHEDULE_H
#define GNNAGG_COMPAT_GRAPH_SCHEDULE_H
#include <vector>

#include "util.h"

enum Schedule { locality, neighbor_grouping, locality_neighbor_grouping, nop };

inline void neighbor_grouping_schedule(int *ptr, int *idx, int neighbor_num, int num_v, int num_e,
                                       std::vector<int> *ptr_vec, std::vector<int> *idx_vec, std::vector<int> *target_vec)
{
    int G = 0;
    checkGnnagg(gnnagg_neighbor_grouping_schedule(ptr, neighbor_num, num_v, nullptr, nullptr, &G));
    ptr_vec->resize((size_t)G + 1);
    target_vec->resize((size_t)G);
    checkGnnagg(gnnagg_neighbor_grouping_schedule(ptr, neighbor_num, num_v, ptr_vec->data(), target_vec->data(), &G));
    idx_vec->assign(idx, idx + num_e);  // reference :121-122: plain copy
}

inline void locality_schedule_impl(int *ptr, int *idx, int par_num, int ng, int num_v, std::vector<int> *ptr_vec,
                                   std::vector<int> *idx_vec, std::vector<int> *target_vec, int total_num_v, float *val,
                                   std::vector<float> *val_vec)
{
    const int E = ptr[num_v];
    ptr_vec->resize((size_t)E + 2);
    idx_vec->resize((size_t)(E > 0 ? E : 1));
    target_vec->resize((size_t)(E > 0 ? E : 1));
    if (val && val_vec) val_vec->resize((size_t)(E > 0 ? E : 1));
    int G = 0;
    checkGnnagg(gnnagg_locality_schedule(ptr, idx, val, par_num, ng, num_v, total_num_v, ptr_vec->data(), idx_vec->data(),
                                         (val && val_vec) ? val_vec->data() : nullptr, target_vec->data(), &G));
    ptr_vec->resize((size_t)G + 1);
    target_vec->resize((size_t)G);
    const int kept = (*ptr_vec)[G];
    idx_vec->resize((size_t)kept);
    if (val && val_vec) val_vec->resize((size_t)kept);
}

inline void locality_schedule(int *ptr, int *idx, int par_num, int num_v, std::vector<int> *ptr_vec,
                              std::vector<int> *idx_vec, std::vector<int> *target_vec, int total_num_v,
                              float *val = nullptr, std::vector<float> *val_vec = nullptr)
{
    locality_schedule_impl(ptr, idx, par_num, 0, num_v, ptr_vec, idx_vec, target_vec, total_num_v, val, val_vec);
}

inline void localityNeighborGrouping(int *ptr, int *idx, int par_num, int neighbor_num, int num_v,
                                     std::vector<int> *ptr_vec, std::vector<int> *idx_vec, std::vector<int> *target_vec,
                                     int total_num_v, float *val = nullptr, std::vector<float> *val_vec = nullptr)
{
    locality_schedule_impl(ptr, idx, par_num, neighbor_num, num_v, ptr_vec, idx_vec, target_vec, total_num_v, val, val_vec);
}
#endif
