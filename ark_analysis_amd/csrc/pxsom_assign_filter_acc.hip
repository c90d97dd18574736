// pxsom_assign_filter_acc.hip -- the batch-rule variant of the register-resident BMU filter (ACC = true in
// pxsom_assign_filter_fast.h): labels + per-BMU sums / counts of a mini-batch in ONE launch, listed rows
// settled inline in binary64.  Compiled WITHOUT -ffinite-math-only: the inline exact path relies on IEEE
// NaN comparisons (a NaN row must end up with label 0).
#include <algorithm>
#include <cstdlib>

#include "pxsom_assign_filter_fast.h"
#include "pxsom_assign_onepass.h"

namespace pxsom_bmu {
namespace {

template <typename T, int CPL, bool FIX, bool FOLD = true>
void launch_acc(const T *x, int64_t n, int c, int64_t ldx, char *ws, const Layout &L, int32_t *labels, double *stats,
                const double *w, hipStream_t st, FinishTables *fin = nullptr)
{
    auto kern = bmu_filter_fast<T, CPL, 7, 1, 0, true, FIX, FOLD>;
    constexpr int kThreads = fast_threads(true, FIX), kWaves = kThreads / 64;   // (FIX: one workgroup of 512 threads per CU)
    // table (first the row-major codebook prep reads) | transposed codebook | fragments | bias | header copy | listed-row queue:
    // 63 KB at k = 100, c = 22 -- two workgroups per CU
    const size_t lds = (fast_table_copies(true, FIX, CPL) * acc_table_words(L.k, c, CPL, FIX) + (PXSOM_FAST_WGS < 3 ? (size_t)L.k * c : 0)) * sizeof(double) +
                       (size_t)7 * 2 * 64 * sizeof(half8) + (size_t)7 * 64 * sizeof(f32x4) + kHdrBytes +
                       256 * sizeof(int64_t) + 16 +   // + queue of listed rows and its counter
                       (size_t)kWaves * 256 * sizeof(int64_t);    // + the waves' queues of rows that wait for the full search
    static pxsom::PerDevice<int> bpc_on;   // (one per instantiation)
    int &bpc = bpc_on.here();
    if (bpc == 0) {
        // (what the largest codebook of the shape needs -- 100 nodes, c = 4 CPL --, not the CU's 160 KB: the kernel has static LDS too)
        const size_t lds_max = (fast_table_copies(true, FIX, CPL) * acc_table_words(100, 4 * CPL, CPL, FIX) + (size_t)100 * 4 * CPL) * sizeof(double) +
                               (size_t)7 * 2 * 64 * sizeof(half8) + (size_t)7 * 64 * sizeof(f32x4) + kHdrBytes + 256 * sizeof(int64_t) + 16 +
                               (size_t)kWaves * 256 * sizeof(int64_t);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
        int nbk = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nbk, kern, kThreads, lds) != hipSuccess || nbk < 1) nbk = 1;
        bpc = nbk > 8 ? 8 : nbk;
    }
    const int64_t ngroups = (n + 63) / 64;
    // two 64-row groups per workgroup when the launch is small (groups are dealt workgroup-major): measured on
    // the 256-group mini-batches of config 2, training pass 1.63 / 1.59 / 1.50 ms for 4 / 1 / 2 groups per workgroup
    // (fewer workgroups share the exact rows among fewer waves; more pay the prologue and the flush more often)
    constexpr int kGroupsPerWg = kWaves / 2;   // (half a group per wave at least)
    int grid = (int)std::min<int64_t>((ngroups + kGroupsPerWg - 1) / kGroupsPerWg, (int64_t)pxsom::device_cu_count() * bpc);
    if (grid < 1) grid = 1;
    // FIX: ceil(log2(rows one workgroup can meet)) -- its waves x 64 rows per round of the grid, shifted last group included
    int fix_rows_log2 = 0;
    {
        const int64_t rows_wg = ((ngroups + (int64_t)grid * kWaves - 1) / ((int64_t)grid * kWaves)) * (64 * kWaves) + 64;
        while (((int64_t)1 << fix_rows_log2) < rows_wg) fix_rows_log2++;
    }
    PXSOM_TIMED_LAUNCH(kern, dim3(grid), dim3(kThreads), lds, st, x, n, c, ldx,
                       reinterpret_cast<const half8 *>(ws + L.off_wfrag),
                       reinterpret_cast<const f32x4 *>(ws + L.off_bias), reinterpret_cast<AssignHdr *>(ws),
                       reinterpret_cast<unsigned *>(ws + L.off_list), labels, L.k, stats, w, L.idx_bits, L.node_bits,
                       fix_rows_log2, (FIX && fin) ? *fin : FinishTables{});
    if (FIX && fin && fin->ticket) fin->done = true;
}

// the one-pass kernel with two tiles per trip (pxsom_assign_onepass.h): fixed-point tables
template <typename T, int CPL, int TMODE = 1>
void launch_onepass(const T *x, int64_t n, int c, int64_t ldx, const Layout &L, int32_t *labels, double *stats, const double *w,
                    hipStream_t st)
{
    constexpr int kThreads = sizeof(T) == 8 ? 512 : kOneThreads, kWpe = sizeof(T) == 8 ? 2 : PXSOM_ONE_WPE, kWaves = kThreads / 64;
    auto kern = bmu_onepass_kernel<T, CPL, kThreads, kWpe, TMODE>;
    const size_t lds = onepass_lds_bytes(L.k, c, kThreads);
    static pxsom::PerDevice<int> bpc_on;   // (one per instantiation)
    int &bpc = bpc_on.here();
    if (bpc == 0) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        int nbk = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nbk, kern, kThreads, lds) != hipSuccess || nbk < 1) nbk = 1;
        bpc = nbk > 4 ? 4 : nbk;
    }
    const int64_t nunits = (n + 31) / 32;
    // at least two 32-row units per wave where the launch is small (fewer workgroups pay the prologue and the flush)
    int grid = (int)std::min<int64_t>((nunits + 2 * kWaves - 1) / (2 * kWaves), (int64_t)pxsom::device_cu_count() * bpc);
    if (grid < 1) grid = 1;
    int fix_rows_log2 = 0;
    {
        const int64_t rows_wg = ((nunits + (int64_t)grid * kWaves - 1) / ((int64_t)grid * kWaves)) * (32 * kWaves) + 32;
        while (((int64_t)1 << fix_rows_log2) < rows_wg) fix_rows_log2++;
    }
    PXSOM_TIMED_LAUNCH(kern, dim3(grid), dim3(kThreads), lds, st, x, n, c, ldx, labels, L.k, stats, w, L.idx_bits, L.node_bits,
                       fix_rows_log2);
}

}  // namespace

template <typename T>
void launch_filter_fast_acc(const T *x, int64_t n, int c, int64_t ldx, char *ws, const Layout &L, int32_t *labels,
                            double *stats, const double *w, hipStream_t st, FinishTables *fin)
{
    const bool fixed = fin != nullptr;
    // (fixed-point tables: the kernel that finds a slot past the row's end to count the rows in, c < 4 CPL, or the one that counts
    // them with an instruction of its own)
#define PXSOM_ACC(CPL)                                                                                           \
    (fixed ? (c < 4 * CPL ? launch_acc<T, CPL, true, true>(x, n, c, ldx, ws, L, labels, stats, w, st, fin)       \
                          : launch_acc<T, CPL, true, false>(x, n, c, ldx, ws, L, labels, stats, w, st, fin))     \
           : launch_acc<T, CPL, false>(x, n, c, ldx, ws, L, labels, stats, w, st))
    // binary64 rows (what the drop-in classes hold) ALWAYS take the two-tile kernel -- fixed-point or binary64 tables --: it is the
    // one without spills (bmu_filter_fast kept four tiles of binary64 rows in flight and spilled 28 - 138 VGPRs; its binary64
    // instantiations are gone).  On binary32 / binary16 rows it was measured slower (0.297 against 0.262 ms, profiles/r05) and is
    // not built for them.
    if constexpr (sizeof(T) == 8) {
#define PXSOM_ONE64(CPL) (fixed ? launch_onepass<T, CPL, 1>(x, n, c, ldx, L, labels, stats, w, st) \
                                : launch_onepass<T, CPL, 2>(x, n, c, ldx, L, labels, stats, w, st))
        if (L.cpl == 8)
            PXSOM_ONE64(8);
        else if (L.cpl == 6)
            PXSOM_ONE64(6);
        else if (L.cpl == 4)
            PXSOM_ONE64(4);
        else
            PXSOM_ONE64(2);
#undef PXSOM_ONE64
        return;
    } else {
    if (L.cpl == 6)
        PXSOM_ACC(6);
    else if (L.cpl == 8)
        PXSOM_ACC(8);
    else if (L.cpl == 4)
        PXSOM_ACC(4);
    else
        PXSOM_ACC(2);
    }
#undef PXSOM_ACC
}

// pxsom_assign on binary64 rows of the register-resident shapes: labels only, in the two-tile kernel (no workspace, no list,
// no exact launch behind it; bmu_filter_fast<double, ACC = false> keeps four tiles of binary64 rows in flight and spills).
bool onepass_labels_route(size_t elem_bytes) { return elem_bytes == 8; }
void launch_onepass_labels(const double *x, int64_t n, int c, int64_t ldx, const Layout &L, int32_t *labels, const double *w,
                           hipStream_t st)
{
    if (L.cpl == 8)
        launch_onepass<double, 8, 0>(x, n, c, ldx, L, labels, nullptr, w, st);
    else if (L.cpl == 6)
        launch_onepass<double, 6, 0>(x, n, c, ldx, L, labels, nullptr, w, st);
    else if (L.cpl == 4)
        launch_onepass<double, 4, 0>(x, n, c, ldx, L, labels, nullptr, w, st);
    else
        launch_onepass<double, 2, 0>(x, n, c, ldx, L, labels, nullptr, w, st);
}

template void launch_filter_fast_acc<float>(const float *, int64_t, int, int64_t, char *, const Layout &, int32_t *,
                                            double *, const double *, hipStream_t, FinishTables *);
template void launch_filter_fast_acc<double>(const double *, int64_t, int, int64_t, char *, const Layout &, int32_t *,
                                             double *, const double *, hipStream_t, FinishTables *);
template void launch_filter_fast_acc<_Float16>(const _Float16 *, int64_t, int, int64_t, char *, const Layout &,
                                               int32_t *, double *, const double *, hipStream_t, FinishTables *);

}  // namespace pxsom_bmu
