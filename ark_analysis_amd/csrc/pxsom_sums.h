// pxsom_sums.h -- per-cluster sums, wave-private tables, two channels per lane (pxsom_sums.hip)
#ifndef PXSOM_SUMS_H
#define PXSOM_SUMS_H
#include <hip/hip_runtime.h>

#include <cstdint>

namespace pxsom {

// sums[label - 1, :] += x[i, :], counts[label - 1] += 1 for rows with an even channel count, an even leading
// dimension and a pair-aligned base (channel PAIRS are loaded: 4 / 8 / 16 bytes for fp16 / fp32 / fp64).  counts_f64: the counts
// buffer holds binary64 (the batch rule's statistics) instead of int64.  nwv tables per workgroup.
// Returns false without launching when the shape is outside this kernel.
template <typename T>
bool launch_sums_pairs(const T *x, int64_t n, int c, int64_t ldx, const int32_t *labels, int k, double *sums,
                       void *counts, bool counts_f64, hipStream_t st, int nwv, int blocks_per_cu);

}  // namespace pxsom
#endif
