// spmm.h -- naive SpMM baseline and the device-side validators with the reference's entry points
// (reference include/spmm.h:35-91, :223-265).
#ifndef GNNAGG_COMPAT_SPMM_H
#define GNNAGG_COMPAT_SPMM_H
#include "util.h"

// reference valid(), spmm.h:35-69: number of elements with |(ref-ans)/ref| > 1e-2
inline int valid(float *y, float *y2, int num)
{
    int diff = 0;
    checkGnnagg(gnnagg_validate(y, y2, num, &diff, nullptr));
    return diff;
}

// reference validReordered(), spmm.h:71-91: ref row r against ans row rows[r] (global `rows`, on the host)
inline int validReordered(float *y, float *y2, int num_v, int flen)
{
    int *d_map = nullptr, diff = 0;
    checkHipErrors(hipMalloc((void **)&d_map, sizeof(int) * num_v));
    checkHipErrors(hipMemcpy(d_map, rows, sizeof(int) * num_v, hipMemcpyHostToDevice));
    checkGnnagg(gnnagg_validate_reordered(y, y2, d_map, num_v, flen, &diff, nullptr));
    (void)hipFree(d_map);
    return diff;
}

// reference spmm<LENFEATURE><<<...>>>, spmm.h:223-265 (feature length is a run-time argument here)
inline void spmm_naive(int numV, int *ptr, int *idx, float *val, float *denseInput, float *denseOutput, int flen)
{
    checkGnnagg(gnnagg_spmm_naive(ptr, idx, val, denseInput, denseOutput, numV, flen, nullptr));
}
#endif
