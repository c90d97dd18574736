// pxsom_batch_step.h -- what the one-launch mini-batch step (pxsom_batch_step.hip) and the persistent BMU-only tail
// (pxsom_batch_tail.hip) share: the grid the register-resident training kernels are built for, and the exact path of a
// listed row whose values sit in LDS.
#pragma once
#include "pxsom_xch.h"
#include <cfloat>
#include <cmath>

#include "pxsom_assign_filter_fast.h"

namespace pxsom_bmu {
namespace {

constexpr int kXD = 10, kYD = 10, kK = 100, kNB = 7;
constexpr int kQueueRows = 96;       // listed rows a workgroup keeps in LDS; further ones are settled on the spot

struct StepHdr {
    int bad;          // NaN / Inf met in the codebook
    unsigned q_n;     // rows in the queue
};

#pragma clang fp contract(off)
// One listed row settled by a whole wave: lanes <-> nodes lane and lane + 64, the row's values read from LDS
// (one address for the wave: a broadcast), distances exactly as the oracle forms them.
// (cs: row stride of the table in words -- c, or c padded to an odd number against LDS bank conflicts: table_stride())
__device__ __forceinline__ void exact_row_from_lds(const double *xr, int c, const double *wt, double *ls, int lane,
                                                   double qmagic, int cs = 0)
{
    if (cs == 0) cs = c;
    const int n0 = lane, n1 = lane + 64;
    const int c1 = n1 < kK ? n1 : kK - 1;
    double d0 = 0.0, d1 = 0.0;
    int j = 0;
    for (; j + 4 <= c; j += 4) {   // the LDS reads of a trip are issued together; sums stay in j order
        double xa[4], wa[4], wb[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            xa[u] = xr[j + u];
            wa[u] = wt[(size_t)(j + u) * kK + n0];
            wb[u] = wt[(size_t)(j + u) * kK + c1];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const double t0 = xa[u] - wa[u], t1 = xa[u] - wb[u];
            d0 += t0 * t0;
            d1 += t1 * t1;
        }
    }
    for (; j < c; j++) {
        const double xj = xr[j];
        const double t0 = xj - wt[(size_t)j * kK + n0], t1 = xj - wt[(size_t)j * kK + c1];
        d0 += t0 * t0;
        d1 += t1 * t1;
    }
    double best = DBL_MAX;
    int bestk = 0x7fffffff;
    const double s0 = sqrt(d0), s1 = sqrt(d1);
    if (s0 < best) {
        best = s0;
        bestk = n0;
    }
    if (n1 < kK && s1 < best) {
        best = s1;
        bestk = n1;
    }
    const double smin = pxsom::wave_min_f64(best);
    const int win = (int)pxsom::wave_min_u32(best == smin ? (unsigned)bestk : 0xffffffffu);
    if (win != 0x7fffffff) {   // 0x7fffffff: no finite distance (NaN row): label 0, not accumulated
        if (lane < c)
            __hip_atomic_fetch_add(ls + (size_t)win * cs + lane, qround(xr[lane], qmagic), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (lane == 0)
            __hip_atomic_fetch_add(ls + (size_t)kK * cs + win, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}
#pragma clang fp contract(fast)

}  // namespace
}  // namespace pxsom_bmu
