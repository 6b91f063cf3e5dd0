// sample.h -- only fullGraph() of the reference's include/sample.h:126-129 is on the aggregation path
// (the GPU samplers are called by no driver).
#ifndef GNNAGG_COMPAT_SAMPLE_H
#define GNNAGG_COMPAT_SAMPLE_H
#include "util.h"
inline CSRSubGraph fullGraph(int *ptr, int *idx) { return CSRSubGraph(nullptr, ptr, idx, n, m); }
#endif
