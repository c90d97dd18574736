// pxsom_assign_filter_acc.hip -- the batch-rule variant of the register-resident BMU filter (ACC = true in
// pxsom_assign_filter_fast.h): labels + per-BMU sums / counts of a mini-batch in ONE launch, listed rows
// settled inline in binary64.  Compiled WITHOUT -ffinite-math-only: the inline exact path relies on IEEE
// NaN comparisons (a NaN row must end up with label 0).
#include <algorithm>

#include "pxsom_assign_filter_fast.h"

namespace pxsom_bmu {
namespace {

template <typename T, int CPL>
void launch_acc(const T *x, int64_t n, int c, int64_t ldx, char *ws, const Layout &L, int32_t *labels, double *stats,
                const double *w, hipStream_t st)
{
    auto kern = bmu_filter_fast<T, CPL, 7, 1, 0, true>;
    // table | transposed codebook | row-major codebook | fragments | bias | header copy | listed-row queue
    const size_t lds = ((size_t)L.k * c + L.k + 2 * (size_t)L.k * c) * sizeof(double) +
                       (size_t)7 * 2 * 64 * sizeof(half8) + (size_t)7 * 64 * sizeof(f32x4) + kHdrBytes +
                       256 * sizeof(int64_t) + 16;   // + queue of listed rows and its counter
    static pxsom::PerDevice<int> bpc_on;
    int &bpc = bpc_on.here();
    if (bpc == 0) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  128 * 1024);
        int nbk = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nbk, kern, 256, lds) != hipSuccess || nbk < 1) nbk = 1;
        bpc = nbk > 8 ? 8 : nbk;
    }
    const int64_t ngroups = (n + 63) / 64;
    // two 64-row groups per workgroup when the launch is small (groups are dealt workgroup-major): measured on
    // the 256-group mini-batches of config 2, training pass 1.63 / 1.59 / 1.50 ms for 4 / 1 / 2 groups per workgroup
    // (fewer workgroups share the exact rows among fewer waves; more pay the prologue and the flush more often)
    int grid = (int)std::min<int64_t>((ngroups + 1) / 2, (int64_t)pxsom::device_cu_count() * bpc);
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, x, n, c, ldx,
                       reinterpret_cast<const half8 *>(ws + L.off_wfrag),
                       reinterpret_cast<const f32x4 *>(ws + L.off_bias), reinterpret_cast<AssignHdr *>(ws),
                       reinterpret_cast<unsigned *>(ws + L.off_list), labels, L.k, stats, w, L.idx_bits, L.node_bits);
}

}  // namespace

template <typename T>
void launch_filter_fast_acc(const T *x, int64_t n, int c, int64_t ldx, char *ws, const Layout &L, int32_t *labels,
                            double *stats, const double *w, hipStream_t st)
{
    if (L.cpl == 6)
        launch_acc<T, 6>(x, n, c, ldx, ws, L, labels, stats, w, st);
    else if (L.cpl == 8)
        launch_acc<T, 8>(x, n, c, ldx, ws, L, labels, stats, w, st);
    else if (L.cpl == 4)
        launch_acc<T, 4>(x, n, c, ldx, ws, L, labels, stats, w, st);
    else
        launch_acc<T, 2>(x, n, c, ldx, ws, L, labels, stats, w, st);
}

template void launch_filter_fast_acc<float>(const float *, int64_t, int, int64_t, char *, const Layout &, int32_t *,
                                            double *, const double *, hipStream_t);
template void launch_filter_fast_acc<double>(const double *, int64_t, int, int64_t, char *, const Layout &, int32_t *,
                                             double *, const double *, hipStream_t);
template void launch_filter_fast_acc<_Float16>(const _Float16 *, int64_t, int, int64_t, char *, const Layout &,
                                               int32_t *, double *, const double *, hipStream_t);

}  // namespace pxsom_bmu
