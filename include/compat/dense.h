// dense.h -- matmul_NN with the reference's signature (reference include/dense.h:4-23), on the library's MFMA kernel.
#ifndef GNNAGG_COMPAT_DENSE_H
#define GNNAGG_COMPAT_DENSE_H
#include "util.h"
// row-major C[inM,inN] = A[inM,inK] . B[inK,inN]; `tmp` (the reference's transpose scratch) is unused.
inline void matmul_NN(float *A, float *B, float *C, int inM, int inN, int inK, float *tmp)
{
    (void)tmp;
    checkGnnagg(gnnagg_matmul_nn(A, B, C, inM, inN, inK, nullptr));
    if (compat_dump_dir()) {   // GNNAGG_COMPAT_DUMP (util.h)
        compat_dump("matmul_NN", "A", A, sizeof(float) * (size_t)inM * inK);
        compat_dump("matmul_NN", "B", B, sizeof(float) * (size_t)inK * inN);
        compat_dump("matmul_NN", "C", C, sizeof(float) * (size_t)inM * inN);
    }
}
#endif
