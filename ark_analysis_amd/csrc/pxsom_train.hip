// pxsom_train.hip -- K6 (SOM training) + K8 (per-cluster sums) on gfx950.
//
//   pxsom_train_online  replaces pyFlowSOM.som (reference call site
//                       /root/reference/src/ark/phenotyping/cluster_helpers.py:106-109): FlowSOM's
//                       C_SOM loop, n*rlen strictly sequential steps.  One persistent workgroup;
//                       thread <-> SOM node; the binary64 codebook lives in LDS ([channel][node], so a
//                       wave's reads are conflict-free); the presented rows are gathered 64 steps
//                       ahead into an LDS ring, so no step waits on HBM; one s_barrier per step.
//                       Latency-bound by construction (neither roofline applies): DESIGN.md "K6a".
//   pxsom_cluster_sums  replaces the pandas groupby-sum of compute_pixel_cluster_channel_avg
//                       (pixel_cluster_utils.py:369-404) and is the accumulation half of the batch rule.
//   pxsom_batch_update  the batch rule's codebook update (oracle of record: orc_batch_update).
#include <algorithm>
#include <cstring>
#include <cfloat>
#include <vector>
#include <cmath>

#include "pxsom_assign.h"
#include "pxsom_common.h"
#include "pxsom_sums.h"
#include "pxsom_wave.h"
#include "pxsom_xch.h"

#ifndef PXSOM_WIDE_WIN_CAP
#define PXSOM_WIDE_WIN_CAP 4096
#endif

namespace {

using pxsom::dpp_f64;
using pxsom::readlane_f64;
using pxsom::shr1_f64;
using pxsom::wave_min_f64;
using pxsom::wave_min_u32;

#pragma clang fp contract(off)

// ------------------------------------------------------------------------------------------------
// exact online SOM.  CMAX > 0: this thread's node (CMAX doubles) and the presented row live in
// registers -- per step one burst of LDS reads for the row, then pure register arithmetic in the
// oracle's order.  CMAX == 0: any channel count, codebook in LDS ([channel][node]) -- or, GLB, where it is:
// codebooks past the LDS (K * C * 8 > ~150 KB, e.g. 264 nodes x 128 channels) are trained in place in w
// (every thread touches only its own node's row; L2-resident, a few microseconds per step).
// ------------------------------------------------------------------------------------------------
// One term of FlowSOM's `change` accumulator (only ever consulted at the start of a pass, when rlen > 1).  The build reads the
// published loop as `change += fabs(tmp)`; PXSOM_ONLINE_INT_ABS is the other recollection -- C's integer abs(), i.e. the
// double truncated towards zero first, which makes every |tmp| < 1 count as 0 (oracle: ORC_V_INT_ABS).
__device__ __forceinline__ double change_term(double tmp, int flags)
{
    return ((flags & PXSOM_ONLINE_INT_ABS) && fabs(tmp) < 2147483648.0) ? (double)abs((int)tmp) : fabs(tmp);
}

template <typename T, int CMAX, int MAXT, bool GLB = false>
__global__ __launch_bounds__(MAXT) void som_online_kernel(const T *__restrict__ x, int64_t n, int c,
                                                          int64_t ldx, double *w, int xdim, int ydim,
                                                          int rlen, double a0, double a1, double r0,
                                                          double r1, const int64_t *__restrict__ order,
                                                          int chunk, int flags)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int K = xdim * ydim;
    const int tid = threadIdx.x, bd = blockDim.x;
    const int lane = tid & 63, wv = tid >> 6, nwv = bd >> 6;
    constexpr bool REG = CMAX > 0;
    static_assert(!(REG && GLB), "register-resident nodes need no codebook storage");
    double *wt = reinterpret_cast<double *>(smem_raw);            // [c][K] (LDS codebook; unused if REG / GLB)
    const int cs = REG ? CMAX : c;                                // row stride of the LDS ring
    double *xs = wt + ((REG || GLB) ? 0 : (size_t)c * K);         // [2][chunk][cs] (pad slots stay 0)
    double *exd = xs + (size_t)2 * chunk * cs;                    // [2][nwv] best distance per wave
    int *exk = reinterpret_cast<int *>(exd + 2 * nwv);            // [2][nwv] best node per wave
    double *red = reinterpret_cast<double *>(exk + 2 * nwv);      // [nwv] change partials
    double *alpha_ring = red + nwv;                               // [2][chunk] learning rate per step

    const bool has_node = tid < K;
    const int node = tid;
    const int nx = node / ydim, ny = node % ydim;
    // channel j of this thread's node, wherever the codebook lives
    auto wref = [&](int j) -> double & { return GLB ? w[(size_t)node * c + j] : wt[(size_t)j * K + node]; };
    double wr[REG ? CMAX : 1];
    if constexpr (REG) {
#pragma unroll
        for (int j = 0; j < CMAX; j++) wr[j] = (has_node && j < c) ? w[(size_t)node * c + j] : 0.0;
    } else if constexpr (!GLB) {
        if (has_node)
            for (int j = 0; j < c; j++) wt[(size_t)j * K + node] = w[(size_t)node * c + j];
    }

    const int64_t niter = (int64_t)rlen * n;
    double threshold = r0;
    const double thresholdStep = (r0 - r1) / (double)niter;
    double change = 1.0;   // uniform: the epoch's total, known after the epoch-boundary reduction
    double mychange = 0.0; // this thread's share of the running epoch
    const bool track = rlen > 1;
    const int per_thread = (chunk * c + bd - 1) / bd;  // gathered elements per thread per chunk
    constexpr int kMaxPer = 16;
    T pre[kMaxPer];  // converted at commit: no use of a loaded value before its chunk is over

    // branch-free per lane (clamped element and step index): a load guarded by a divergent branch
    // gets its s_waitcnt right behind it, which serialises the HBM round trips
    auto gather = [&](int64_t step0) {
#pragma unroll
        for (int u = 0; u < kMaxPer; u++) {
            if (u < per_thread) {  // uniform
                const int e = min(tid + u * bd, chunk * c - 1);
                const int s = e / c, j = e - s * c;
                const int64_t st = step0 + s < niter ? step0 + s : niter - 1;
                pre[u] = x[order[st] * ldx + j];
            }
        }
    };
    // the oracle's alpha = a0 - (a0 - a1) * k / niter (same operation order), one lane per step,
    // so the binary64 division is off the per-step critical path
    auto alphas = [&](int buf, int64_t step0) {
        if (tid < chunk) {
            const int64_t kk = step0 + tid;
            alpha_ring[buf * chunk + tid] = a0 - (a0 - a1) * (double)kk / (double)niter;
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int u = 0; u < kMaxPer; u++) {
            const int e = tid + u * bd;
            if (u < per_thread && e < chunk * c) {
                const int srow = e / c, j = e - srow * c;
                xs[((size_t)buf * chunk + srow) * cs + j] = (double)pre[u];
            }
        }
    };

    // zero the ring once: slots j >= c of every row are never written again, so the unguarded
    // CMAX-long register loops below add exact zeros (x + 0 == x: bit-exactness is preserved)
    for (int e = tid; e < 2 * chunk * cs; e += bd) xs[e] = 0.0;
    __syncthreads();
    gather(0);
    commit(0);
    alphas(0, 0);
    __syncthreads();

    bool done = false;
    int par = 0;
    int64_t in_epoch = 0;  // step % n without a 64-bit division per step
    int cur_buf = 0;
    for (int64_t step0 = 0; step0 < niter && !done; step0 += chunk) {
        const int buf = cur_buf;
        gather(step0 + chunk);  // in flight while this chunk computes
        const double *xc = xs + (size_t)buf * chunk * cs;
        for (int s = 0; s < chunk; s++) {
            const int64_t step = step0 + s;
            if (step >= niter) break;
            int64_t k = step;
            const bool epoch_start = in_epoch == 0;
            if (++in_epoch == n) in_epoch = 0;
            if (epoch_start) {
                if (step > 0) {
                    // epoch boundary: total |delta| of the finished epoch (summation order differs
                    // from the oracle's sequential one; only `change < 1` is ever looked at)
                    double v = mychange;
                    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
                    if (lane == 0) red[wv] = v;
                    __syncthreads();
                    change = 0.0;
                    for (int i = 0; i < nwv; i++) change += red[i];
                    __syncthreads();
                }
                if (change < 1.0) {
                    k = niter;  // FlowSOM: body runs once more with k == niter, then the loop ends
                    done = true;
                }
                change = 0.0;
                mychange = 0.0;
            }
            const double *xr = xc + (size_t)s * cs;
            // squared distance of this thread's node (binary64, j ascending) -- FlowSOM eucl() before
            // its sqrt.  The oracle compares sqrt(d2) values with a strict '<' (first minimum wins).
            // sqrt is monotone, so whenever the smallest d2 is isolated by more than a few ulps its
            // node is the answer and no sqrt is evaluated; only near-coincident candidates (d2 within
            // 2^-50 relative of the minimum) take the sqrt path, which reproduces the oracle's ties.
            double xreg[REG ? CMAX : 1];
            double d2 = INFINITY;
            if constexpr (REG) {
#pragma unroll
                for (int j = 0; j < CMAX; j++) xreg[j] = xr[j];  // one burst of broadcast LDS reads
                double xdist = 0.0;
#pragma unroll
                for (int j = 0; j < CMAX; j++) {
                    const double tmp = xreg[j] - wr[j];  // pad slots: 0 - 0
                    xdist += tmp * tmp;
                }
                if (has_node && xdist == xdist) d2 = xdist;
            } else if (has_node) {
                double xdist = 0.0;
                for (int j = 0; j < c; j++) {
                    const double tmp = xr[j] - wref(j);
                    xdist += tmp * tmp;
                }
                if (xdist == xdist) d2 = xdist;
            }
            const double near_eps = 8.881784197001252e-16;  // 2^-50
            double wmin = wave_min_f64(d2);
            unsigned long long cand = __ballot(d2 <= wmin + wmin * near_eps);
            int bk;
            double bd2;
            if (__popcll(cand) == 1) {
                bk = (wv << 6) + (int)__ffsll((long long)cand) - 1;
                bd2 = wmin;
            } else {
                const double sd = sqrt(d2);
                const double smin = wave_min_f64(sd);
                cand = __ballot(sd == smin);
                const int first = cand ? (int)__ffsll((long long)cand) - 1 : 0;
                bk = (wv << 6) + first;
                bd2 = __shfl(d2, first);
            }
            if (bk >= K) bk = 0x7fffffff;  // only padding lanes (all-infinite wave)
            int nearest = bk;
            if (nwv > 1) {
                if (lane == 0) {
                    exd[par * nwv + wv] = bd2;
                    exk[par * nwv + wv] = bk;
                }
                __syncthreads();
                double gmin = exd[par * nwv];
                for (int i = 1; i < nwv; i++) gmin = fmin(gmin, exd[par * nwv + i]);
                const double lim = gmin + gmin * near_eps;
                int ncand = 0;
                for (int i = 0; i < nwv; i++) {
                    if (exd[par * nwv + i] <= lim) {
                        if (ncand == 0) nearest = exk[par * nwv + i];
                        ncand++;
                    }
                }
                if (ncand > 1) {  // near-coincident minima in different waves: compare like the oracle
                    double best = INFINITY;
                    nearest = 0x7fffffff;
                    for (int i = 0; i < nwv; i++) {
                        const double sdi = sqrt(exd[par * nwv + i]);
                        const int ki = exk[par * nwv + i];
                        if (sdi < best || (sdi == best && ki < nearest)) {
                            best = sdi;
                            nearest = ki;
                        }
                    }
                }
                par ^= 1;
            }
            if (nearest >= K) nearest = 0;
            if (threshold < 1.0) threshold = 0.5;
            const double alpha = k == step ? alpha_ring[buf * chunk + s]
                                           : a0 - (a0 - a1) * (double)k / (double)niter;  // early-stop step
            if (has_node) {
                const int bx = nearest / ydim, by = nearest % ydim;
                const int dx = nx > bx ? nx - bx : bx - nx, dy = ny > by ? ny - by : by - ny;
                const double nh = (double)(dx > dy ? dx : dy);
                if (!(nh > threshold)) {
                    if constexpr (REG) {
#pragma unroll
                        for (int j = 0; j < CMAX; j++) {
                            const double tmp = xreg[j] - wr[j];
                            if (track) mychange += change_term(tmp, flags);  // only ever consulted when rlen > 1
                            wr[j] = wr[j] + tmp * alpha;
                        }
                    } else {
                        for (int j = 0; j < c; j++) {
                            const double wv_ = wref(j);
                            const double tmp = xr[j] - wv_;
                            mychange += change_term(tmp, flags);
                            wref(j) = wv_ + tmp * alpha;
                        }
                    }
                }
            }
            threshold -= thresholdStep;
            if (done) break;
        }
        // publish the next chunk's rows (other buffer: nobody reads it during this chunk)
        commit(buf ^ 1);
        alphas(buf ^ 1, step0 + chunk);
        cur_buf ^= 1;
        __syncthreads();
    }
    if (has_node) {
        if constexpr (REG) {
#pragma unroll
            for (int j = 0; j < CMAX; j++)
                if (j < c) w[(size_t)node * c + j] = wr[j];
        } else if constexpr (!GLB) {
            for (int j = 0; j < c; j++) w[(size_t)node * c + j] = wt[(size_t)j * K + node];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// exact online SOM, split form: L adjacent lanes share one node, each owning CH consecutive channels
// (K * L <= 256 threads: the Pixie default 10x10 x <= 24 markers runs as 4 waves, one per SIMD, 12
// channels per lane).  binary64 issue (~6.75 cycles per wave instruction) bounds the thread<->node
// form; splitting the channels cuts the per-wave instruction count.  The codebook update is
// element-wise and splits trivially.  The winner search runs in two tiers, both bit-faithful:
//  * every step: squared distances by a pairwise tree + butterfly (short dependency chain), wave minimum
//    on their upper 32 bits (one v_min_u32 DPP per stage).  The tree sum is within a few ulp of the
//    oracle's left-to-right sum, so it decides the step only when no other node's key is within one key
//    step (2^-21 relative) of the leader's -- then the oracle's winner is necessarily the same node;
//  * otherwise (near ties, duplicates, non-finite rows; block-uniform branch): the oracle's own
//    arithmetic -- lane q continues the strictly left-to-right partial sum of lane q-1 (row_shr:1 DPP),
//    so after L phases lane L-1 holds the oracle's value bit for bit -- a second exchange, and the
//    key / sqrt comparison of the thread<->node form.
// ------------------------------------------------------------------------------------------------
// scripts/ubench/online_step_timing.hip includes this file with PXSOM_STEP_TIMING defined: s_memtime
// deltas per step segment, accumulated by wave 0 (changes the schedule slightly; diagnosis only)
#ifdef PXSOM_STEP_TIMING
__device__ long long g_step_ticks[8];
#define PXSOM_TICK(i)                                 \
    do {                                              \
        const long long t_now = clock64();            \
        tick_acc[i] += t_now - tick_prev;             \
        tick_prev = t_now;                            \
    } while (0)
#else
#define PXSOM_TICK(i) \
    do {              \
    } while (0)
#endif

template <typename T, int CH, int L>
__global__ __launch_bounds__(CH * L > 40 ? 512 : 256) void som_online_split_kernel(const T *__restrict__ x, int64_t n, int c,
                                                               int64_t ldx, double *w, int xdim, int ydim,
                                                               int rlen, double a0, double a1, double r0,
                                                               double r1, const int64_t *__restrict__ order,
                                                               int chunk, int flags)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int CMAX = CH * L;
    constexpr int LOG_L = L == 4 ? 2 : 1;
    constexpr int NS = 128;  // distance slots per parity (K <= 128); slots >= K stay +inf
    const int K = xdim * ydim;
    const int tid = threadIdx.x, bd = blockDim.x;
    const int lane = tid & 63, wv = tid >> 6, nwv = bd >> 6;
    double *xs = reinterpret_cast<double *>(smem_raw);        // [2][chunk][CMAX] (pad slots stay 0)
    double *dall = xs + (size_t)2 * chunk * CMAX;             // [2][NS] squared distance of every node
    double *dexact = dall + 2 * NS;                           // [NS] FlowSOM-order distances of a close call
    double *red = dexact + NS;                                // [8] change partials (one per wave)
    int64_t *ordl = reinterpret_cast<int64_t *>(red + 8);     // [chunk] rows presented in the next chunk
    double *alpha_ring = reinterpret_cast<double *>(ordl + chunk);  // [2][chunk] (+ CMAX doubles of slack:
                                                                    //  the one-step-ahead reads may overrun)

    const int node = tid >> LOG_L, q = tid & (L - 1);
    const bool has_node = node < K;
    const bool owner = has_node && q == L - 1;                // this lane ends up with the node's d2
    const int nx = node / ydim, ny = node % ydim;
    // every wave searches all K distances after the exchange: lane l looks at nodes 2l and 2l + 1
    const int pk0 = (2 * lane) | (((2 * lane) / ydim) << 8) | (((2 * lane) % ydim) << 16);
    const int pk1 = (2 * lane + 1) | (((2 * lane + 1) / ydim) << 8) | (((2 * lane + 1) % ydim) << 16);
    const int ch0 = q * CH;
    double wr[CH];
#pragma unroll
    for (int j = 0; j < CH; j++) wr[j] = (has_node && ch0 + j < c) ? w[(size_t)node * c + ch0 + j] : 0.0;

    const int64_t niter = (int64_t)rlen * n;
    double threshold = r0;
    const double thresholdStep = (r0 - r1) / (double)niter;
    double change = 1.0, mychange = 0.0;
    const bool track = rlen > 1;
    const int per_thread = (chunk * c + bd - 1) / bd;
    constexpr int kMaxPer = 16;
    T pre[kMaxPer];
    int64_t ord_pre = 0;

    // presented rows travel HBM -> registers -> LDS one chunk ahead of their use; their row numbers
    // (order[]) two chunks ahead, so neither dependent HBM round trip is ever waited on mid-chunk
    auto fetch_order = [&](int64_t step0) {  // branch-free as well (steps past the end re-read the last)
        const int64_t st = step0 + min(tid, chunk - 1);
        ord_pre = order[st < niter ? st : niter - 1];
    };
    auto publish_order = [&]() {
        if (tid < chunk) ordl[tid] = ord_pre;
    };
    // branch-free per lane (clamped element index, row 0 past the end): a load guarded by a divergent
    // branch gets its s_waitcnt right behind it, which serialises the HBM round trips
    auto gather = [&](int64_t) {
#pragma unroll
        for (int u = 0; u < kMaxPer; u++) {
            if (u < per_thread) {  // uniform
                const int e = min(tid + u * bd, chunk * c - 1);
                const int s = e / c, j = e - s * c;
                pre[u] = x[ordl[s] * ldx + j];
            }
        }
    };
    auto commit = [&](int buf, int64_t step0) {
#pragma unroll
        for (int u = 0; u < kMaxPer; u++) {
            const int e = tid + u * bd;
            if (u < per_thread && e < chunk * c) {
                const int srow = e / c, j = e - srow * c;
                xs[((size_t)buf * chunk + srow) * CMAX + j] = (double)pre[u];
            }
        }
        if (tid < chunk) {
            const int64_t kk = step0 + tid;
            alpha_ring[buf * chunk + tid] = a0 - (a0 - a1) * (double)kk / (double)niter;
        }
    };

    for (int e = tid; e < 2 * chunk * CMAX; e += bd) xs[e] = 0.0;
    for (int e = tid; e < 3 * NS; e += bd) dall[e] = INFINITY;  // dall and dexact
    fetch_order(0);
    publish_order();
    __syncthreads();
    gather(0);
    fetch_order(chunk);
    commit(0, 0);
    __syncthreads();  // everyone has consumed ordl (the loads above have returned)
    publish_order();
    __syncthreads();
    // nothing issued so far is still in flight: without this the s_waitcnt pass keeps a vmcnt(0) at
    // the top of the step loop, which would wait for every chunk's prefetch
    __builtin_amdgcn_s_waitcnt(0x0F70);

    bool done = false;
    int par = 0, buf = 0;
    int64_t in_epoch = 0;
    typedef double d2_t __attribute__((ext_vector_type(2)));
#ifdef PXSOM_STEP_TIMING
    long long tick_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tick_prev = clock64();
#endif
    for (int64_t step0 = 0; step0 < niter && !done; step0 += chunk) {
        fetch_order(step0 + 2 * chunk);  // row numbers of the chunk after the next
        gather(step0 + chunk);           // rows of the next chunk (their numbers are in ordl)
        const double *xc = xs + (size_t)buf * chunk * CMAX + ch0;
        PXSOM_TICK(7);
        // the row and learning rate of step s + 1 are read from LDS while step s computes
        double xcur[CH], alpha_cur = alpha_ring[buf * chunk];
#pragma unroll
        for (int j = 0; j < CH; j++) xcur[j] = xc[j];
        for (int s = 0; s < chunk; s++) {
            const int64_t step = step0 + s;
            if (step >= niter) break;
            int64_t k = step;
            const bool epoch_start = in_epoch == 0;
            if (++in_epoch == n) in_epoch = 0;
            if (epoch_start) {
                if (step > 0) {
                    double v = mychange;
                    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
                    if (lane == 0) red[wv] = v;
                    __syncthreads();
                    change = 0.0;
                    for (int i = 0; i < nwv; i++) change += red[i];
                    __syncthreads();
                }
                if (change < 1.0) {
                    k = niter;
                    done = true;
                }
                change = 0.0;
                mychange = 0.0;
            }
            PXSOM_TICK(0);
            double tmp[CH], sq[CH];
#pragma unroll
            for (int j = 0; j < CH; j++) {
                tmp[j] = xcur[j] - wr[j];  // pad slots: 0 - 0
                sq[j] = tmp[j] * tmp[j];
            }
            const double alpha_ring_s = alpha_cur;
            {
                const double *xr = xc + (size_t)(s + 1) * CMAX;  // s + 1 == chunk: in-bounds, unused
#pragma unroll
                for (int j = 0; j < CH; j++) xcur[j] = xr[j];
                alpha_cur = alpha_ring[buf * chunk + s + 1];
            }
            // Squared distance, fast form: pairwise tree over the lane's channels, butterfly over the node's
            // L lanes.  It differs from FlowSOM's left-to-right sum by a few ulp at most (< 2^-47 relative),
            // so it may only decide a step whose runner-up is far away; see the key test below.
            double tsum[CH];
#pragma unroll
            for (int j = 0; j < CH; j++) tsum[j] = sq[j];
#pragma unroll
            for (int stride = 1; stride < CH; stride *= 2)
#pragma unroll
                for (int j = 0; j + stride < CH; j += 2 * stride) tsum[j] += tsum[j + stride];
            double dfast = tsum[0];
            dfast += dpp_f64(dfast, 0);
            if (L == 4) dfast += dpp_f64(dfast, 1);
            PXSOM_TICK(1);
            // all-to-all through LDS: one write, one barrier, one 16-byte read per lane; every wave then
            // finds the minimum of all K distances itself (no second exchange of per-wave winners)
            if (owner) dall[par * NS + node] = dfast == dfast ? dfast : INFINITY;
            __syncthreads();
            PXSOM_TICK(2);
            const d2_t df = *reinterpret_cast<const d2_t *>(dall + par * NS + 2 * lane);
            par ^= 1;
            // keys = upper 32 bits of d2 (sign, exponent, 20 mantissa bits).  Exactly one key within
            // {kmin, kmin + 1}: every other node is at least one whole key step (>= 2^-21 relative) above the
            // leader -- far more than the fast sum can be off -- so the leader is FlowSOM's winner as well.
            const unsigned fk0 = (unsigned)(__double_as_longlong(df[0]) >> 32);
            const unsigned fk1 = (unsigned)(__double_as_longlong(df[1]) >> 32);
            const unsigned fkmin = wave_min_u32(min(fk0, fk1));
            const unsigned long long near0 = __ballot(fk0 <= fkmin + 1u), near1 = __ballot(fk1 <= fkmin + 1u);
            int nearest;
            if (__popcll(near0) + __popcll(near1) == 1 && fkmin < 0x7ff00000u) {
                nearest = near0 ? __builtin_amdgcn_readlane(pk0, (int)__ffsll((long long)near0) - 1)
                                : __builtin_amdgcn_readlane(pk1, (int)__ffsll((long long)near1) - 1);
            } else {
                // close call (or no finite distance): redo the step with FlowSOM's own arithmetic.
                // eucl() before its sqrt: xdist = 0; xdist += tmp_j^2, j ascending over the node's channels
                // (0 + sq_0 == sq_0 exactly); lane q continues the partial sum of lane q - 1.
                double acc = sq[0];
#pragma unroll
                for (int j = 1; j < CH; j++) acc += sq[j];
#pragma unroll
                for (int p = 1; p < L; p++) {
                    double t = shr1_f64(acc);
#pragma unroll
                    for (int j = 0; j < CH; j++) t += sq[j];
                    acc = q == 0 ? acc : t;
                }
                if (owner) dexact[node] = acc == acc ? acc : INFINITY;
                __syncthreads();  // block-uniform branch: every wave read the same distances
                const d2_t dd = *reinterpret_cast<const d2_t *>(dexact + 2 * lane);
                // the oracle compares sqrt(d2) with a strict '<' in node order.  sqrt is monotone: a node
                // whose d2 is the only one with the smallest key wins outright; keys shared by several nodes
                // are settled on the sqrt values themselves.
                const unsigned key0 = (unsigned)(__double_as_longlong(dd[0]) >> 32);
                const unsigned key1 = (unsigned)(__double_as_longlong(dd[1]) >> 32);
                const unsigned kmin = wave_min_u32(min(key0, key1));
                unsigned long long cand0 = __ballot(key0 == kmin), cand1 = __ballot(key1 == kmin);
                if (__popcll(cand0) + __popcll(cand1) == 1) {
                    nearest = cand0 ? __builtin_amdgcn_readlane(pk0, (int)__ffsll((long long)cand0) - 1)
                                    : __builtin_amdgcn_readlane(pk1, (int)__ffsll((long long)cand1) - 1);
                } else {
                    const double s0 = key0 == kmin ? sqrt(dd[0]) : INFINITY;
                    const double s1 = key1 == kmin ? sqrt(dd[1]) : INFINITY;
                    const double sl = fmin(s0, s1);
                    const double smin = wave_min_f64(sl);
                    const unsigned long long cl = __ballot(sl == smin);
                    const int first = (int)__ffsll((long long)cl) - 1;  // lanes ascend in node order
                    const int p0 = __builtin_amdgcn_readlane(pk0, first), p1 = __builtin_amdgcn_readlane(pk1, first);
                    const bool zero_first = (__ballot(s0 == smin) >> first) & 1ull;
                    nearest = zero_first ? p0 : p1;
                    // no finite distance anywhere (NaN row, overflow): FlowSOM's loop never replaces node 0
                    if (!(smin < INFINITY)) nearest = 0;
                }
            }
            PXSOM_TICK(4);
            if (threshold < 1.0) threshold = 0.5;
            const double alpha = k == step ? alpha_ring_s : a0 - (a0 - a1) * (double)k / (double)niter;
            const int bx = (nearest >> 8) & 0xff, by = (nearest >> 16) & 0xff;
            const int dx = nx > bx ? nx - bx : bx - nx, dy = ny > by ? ny - by : by - ny;
            const double nh = (double)(dx > dy ? dx : dy);
            if (has_node && !(nh > threshold)) {
#pragma unroll
                for (int j = 0; j < CH; j++) wr[j] = wr[j] + tmp[j] * alpha;
                if (track) {  // only ever consulted when rlen > 1
#pragma unroll
                    for (int j = 0; j < CH; j++) mychange += change_term(tmp[j], flags);
                }
            }
            threshold -= thresholdStep;
            PXSOM_TICK(5);
            if (done) break;
        }
        commit(buf ^ 1, step0 + chunk);
        publish_order();
        buf ^= 1;
        __syncthreads();
        PXSOM_TICK(6);
    }
#ifdef PXSOM_STEP_TIMING
    if (tid == 0)
        for (int i = 0; i < 8; i++) g_step_ticks[i] = tick_acc[i];
#endif
    if (has_node) {
#pragma unroll
        for (int j = 0; j < CH; j++)
            if (ch0 + j < c) w[(size_t)node * c + ch0 + j] = wr[j];
    }
}

// ------------------------------------------------------------------------------------------------
// batch update: one workgroup per node k, thread <-> channel.  Only the Chebyshev window of k is
// visited, in the oracle's separable summation order (orc_batch_update: per grid row, then over the rows).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void batch_update_kernel(double *w, int xdim, int ydim, int c,
                                                           const double *__restrict__ sums,
                                                           const double *__restrict__ counts,
                                                           double thr, double q, double sat, int stage,
                                                           double *__restrict__ zero_out, int zero_count,
                                                           const double *w_src = nullptr)
{
    if (!w_src) w_src = w;   // (w_src != w: the updated codebook goes to w, the source stays as it is -- no copy launch)
    extern __shared__ __attribute__((aligned(16))) char upd_smem[];
    const int k = blockIdx.x, tid = threadIdx.x;
    // the OTHER statistics buffer (the next accumulate's target) is cleared here, a slice per workgroup
    if (zero_count > 0) {
        const int per = (zero_count + gridDim.x - 1) / gridDim.x;
        for (int e = k * per + tid; e < min((k + 1) * per, zero_count); e += 256) zero_out[e] = 0.0;
    }
    const int kx = k / ydim, ky = k % ydim;
    // nodes b with max(|dx|, |dy|) <= thr  <=>  |dx|, |dy| <= floor(thr)   (integer distances)
    const int r = thr < 0.0 ? -1 : (thr > 1.0e6 ? 1000000 : (int)floor(thr));
    const int x0 = kx - r < 0 ? 0 : kx - r, x1 = kx + r > xdim - 1 ? xdim - 1 : kx + r;
    const int y0 = ky - r < 0 ? 0 : ky - r, y1 = ky + r > ydim - 1 ? ydim - 1 : ky + r;
    // The window rows x0..x1 are contiguous in node order: stage their statistics in LDS with every
    // thread of the workgroup loading (8 in flight each) -- a loop of dependent L2 round trips per window
    // node costs 5-10 us at radius 6 -- then sum from LDS in the oracle's order.
    const int b_lo = x0 * ydim, b_hi = (x1 + 1) * ydim;  // node range [b_lo, b_hi)
    const double *ls = sums, *lc = counts;
    int boff = 0;
    if (stage) {
        double *ss = reinterpret_cast<double *>(upd_smem);  // [(b_hi - b_lo) * c] sums, then counts
        const int ne = (b_hi - b_lo) * c, nc = b_hi - b_lo;
        const double *gs = sums + (size_t)b_lo * c, *gc = counts + b_lo;
        for (int e0 = tid; e0 < ne + nc; e0 += 8 * 256) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int e = e0 + u * 256;
                v[u] = e < ne ? gs[e] : (e < ne + nc ? gc[e - ne] : 0.0);
            }
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (e0 + u * 256 < ne + nc) ss[e0 + u * 256] = v[u];
        }
        __syncthreads();
        ls = ss;
        lc = ss + ne;
        boff = b_lo;
    }
    for (int j = tid; j < c; j += 256) {   // (wide rows: more channels than threads)
        const double wv = w_src[(size_t)k * c + j];
        // separable order of orc_batch_update: T[bx] = sum over the window's by (ascending), num = sum of T[bx]
        double num = 0.0, den = 0.0;
        for (int bx = x0; bx <= x1; bx++) {
            double tn = 0.0, td = 0.0;
#pragma unroll 4
            for (int by = y0; by <= y1; by++) {
                const int b = bx * ydim + by - boff;
                td += lc[b];
                tn += ls[(size_t)b * c + j];
            }
            den += td;
            num += tn;
        }
        if (den > 0.0) {
            // 1 - (1-alpha)^den by binary exponentiation, 1 - alpha formed on the host (batch_gain; orc_batch_update)
            const double gain = pxsom_bmu::batch_gain(den, q, sat), inv = 1.0 / den;
            // gain == 1 exactly (wide windows): the node is the window mean itself, so nodes sharing a window are
            // bit-identical (and masked as duplicates by prep) instead of one ulp apart (orc_batch_update)
            w[(size_t)k * c + j] = gain == 1.0 ? num * inv : wv + gain * (num * inv - wv);
        } else if (w_src != w) {
            w[(size_t)k * c + j] = wv;
        }
    }
}

#pragma clang fp contract(fast)

// ------------------------------------------------------------------------------------------------
// per-cluster sums/counts.  Each workgroup owns a contiguous row range and a private binary64
// table in LDS (ds_add_f64), flushed once with global_atomic_add_f64.
// Loads are flat-coalesced: lane e reads element e of the row range.
// ------------------------------------------------------------------------------------------------
// How a value joins the workgroup's table.  binary32 / binary64 rows: ds_add_f64.  binary16 rows: every binary16 number is
// an integer multiple of 2^-24 below 2^16, so v * 2^24 is an integer below 2^40 and the table holds exact 64-bit
// fixed-point sums (ds_add_u64: 5.8 lane-atomics per clock per CU against 3.0 for ds_add_f64,
// scripts/ubench/lds_atomic_rate.hip -- this kernel is bound by that rate).  Exact and order-independent: equal to the
// oracle's binary64 sum whenever that one is exact too (sums below 2^29).  Infinities / NaNs go straight to the global
// binary64 table, where they poison the sum as they do in the oracle.
template <typename T>
struct TableAdd {
    static constexpr bool kFixed = false;
    static __device__ __forceinline__ void add(double *ls, size_t slot, T v, double *, size_t = 0)
    {
        __hip_atomic_fetch_add(ls + slot, (double)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    static __device__ __forceinline__ double value(const double *ls, size_t slot) { return ls[slot]; }
};
template <>
struct TableAdd<_Float16> {
    static constexpr bool kFixed = true;
    static __device__ __forceinline__ void add(double *ls, size_t slot, _Float16 v, double *global_sums, size_t global_slot)
    {
        const unsigned b = __builtin_bit_cast(unsigned short, v);
        const unsigned e = (b >> 10) & 31u, m = b & 1023u;
        if (e == 31u) {   // inf / NaN
            __hip_atomic_fetch_add(global_sums + global_slot, (double)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        const unsigned long long q = e ? (unsigned long long)(1024u + m) << (e - 1u) : (unsigned long long)m;
        const unsigned long long sq = (b & 0x8000u) ? 0ull - q : q;   // two's complement
        if (sq)
            __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(ls) + slot, sq, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    static __device__ __forceinline__ double value(const double *ls, size_t slot)
    {
        return (double)reinterpret_cast<const long long *>(ls)[slot] * 0x1p-24;
    }
};

constexpr int kSumsSpare = 16;   // table slots behind the last cluster: where elements without a valid label go

template <typename T, bool COUNT_F64, int NT>
__global__ __launch_bounds__(NT) void cluster_sums_kernel(const T *__restrict__ x, int64_t n, int c,
                                                           int64_t ldx, const int32_t *__restrict__ labels,
                                                           int k, double *sums, unsigned long long *counts,
                                                           int64_t rows_per_block, int use_lds, double qmagic, int cs, pxsom::RowView rv)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    // table row stride cs (words): c, or c padded to an odd number -- the lanes of a ds_add hit rows of unrelated labels, and with
    // an even stride those fall on a fraction of the banks (c = 40: stride 80 dwords, four bank groups in all)
    double *ls = reinterpret_cast<double *>(smem_raw);                 // [k*cs] + kSumsSpare slots nobody reads
    unsigned *lc = reinterpret_cast<unsigned *>(ls + (size_t)k * cs + kSumsSpare);   // [k]
    const int tid = threadIdx.x;
    if (use_lds) {
        for (int e = tid; e < k * cs + kSumsSpare; e += NT) ls[e] = 0.0;
        for (int e = tid; e < k; e += NT) lc[e] = 0u;
        __syncthreads();
    }
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > n) r1 = n;
    // contiguous fp32 rows (ldx == c): the row range is one flat array -- 16-byte loads, 4 per thread in
    // flight (a dword per lane keeps too few bytes in flight for HBM: measured 1.8 TB/s at 10 M rows)
    bool done = false;
    if constexpr (sizeof(T) <= 4) {
        // contiguous fp32 / fp16 rows: VEC = 16 / sizeof(T) elements per load
        constexpr int VEC = 16 / (int)sizeof(T);
        // (a scheduled step's rows where they lie, pxsom::RowView: rows of whole vectors only -- the host sees to it --, a vector's
        // address from its row's place in the caller's matrix instead of from the flat range)
        const bool viewed = rv.gw > 1;
        if (use_lds && ldx == c && r0 < r1 &&
            ((reinterpret_cast<uintptr_t>(x) + (viewed ? (size_t)0 : (size_t)r0 * c * sizeof(T))) & 15) == 0) {
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
            const T *xb = x + r0 * c;
            const int64_t total = (r1 - r0) * c, nvec = total / VEC;
            // (row, channel) of a thread's vector advance by NT*VEC elements per load: no division in the loop
            int64_t vrow = ((int64_t)VEC * tid) / c;
            int vch = (int)((int64_t)VEC * tid - vrow * c);
            const int drow = (NT * VEC) / c, dch = (NT * VEC) % c;
            if (c >= VEC) {
                // A 16-byte vector spans at most two rows: both labels are requested up front, unconditionally (clamped
                // row), and every element picks its own -- no load behind a divergent branch (such a load gets its
                // s_waitcnt right behind it: one serialised L2 round trip per vector, which held this kernel at a
                // quarter of the LDS atomic rate)
                const int64_t last_row = r1 - r0 - 1;
                const bool rows_of_vectors = c % VEC == 0;   // (wave-uniform)
                for (int64_t v0 = tid; v0 < nvec; v0 += 4 * NT) {
                    T val[4][VEC];
                    int lab_a[4], lab_b[4], ch0[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int64_t v = v0 + u * NT;
                        const bool ok = v < nvec;
                        const int64_t row = vrow < last_row ? vrow : last_row, row2 = vrow + 1 < last_row ? vrow + 1 : last_row;
                        const u4 raw = *reinterpret_cast<const u4 *>(viewed ? x + rv.offset(r0 + row, c) + (ok ? vch : 0) : xb + VEC * (ok ? v : nvec - 1));
                        __builtin_memcpy(val[u], &raw, 16);
                        // (rows of whole vectors never look at the second label: its load is the third of every four vector-memory
                        // instructions of this loop)
                        const int la = labels[r0 + row] - 1, lb2 = rows_of_vectors ? -1 : labels[r0 + row2] - 1;
                        lab_a[u] = ok ? la : -1;
                        lab_b[u] = (ok && vrow + 1 <= last_row) ? lb2 : -1;
                        ch0[u] = vch;
                        vrow += drow;
                        vch += dch;
                        if (vch >= c) {
                            vch -= c;
                            vrow++;
                        }
                    }
                    // Round 5: rows of a whole number of vectors (c % VEC == 0: 40 binary16 channels, 100 binary32 columns): a vector lies
                    // in ONE row, so its eight (four) adds go to consecutive words of one table row -- no per-element choice between two
                    // labels, no test for zero (a zero adds nothing), the element's place in the instruction's offset field.  Timing builds
                    // had shown what bounds this kernel: not HBM (0.353 ms of 0.402 without its loads), not the LDS atomics (0.364 without
                    // them), but the dozen vector instructions per ELEMENT around them (profiles/r05/sums_loads_in_flight.txt).
                    if (rows_of_vectors) {
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            const bool ok_a = (unsigned)lab_a[u] < (unsigned)k;
                            const int base = ok_a ? lab_a[u] * cs + ch0[u] : k * cs;
                            bool plain = true;
                            if constexpr (TableAdd<T>::kFixed) {
                                unsigned raw[4], reach = 0u;
                                __builtin_memcpy(raw, val[u], 16);
#pragma unroll
                                for (int d = 0; d < 4; d++) reach |= (raw[d] & 0x7fff7fffu) + 0x04000400u;
                                plain = (reach & 0x80008000u) == 0u;   // (no Inf / NaN among the eight)
                                if (plain) {
                                    unsigned long long *tp = reinterpret_cast<unsigned long long *>(ls) + base;
#pragma unroll
                                    for (int i = 0; i < VEC; i++) {
                                        const double shifted = (double)val[u][i] + 0x1.8p+28;
                                        __hip_atomic_fetch_add(tp + i, (unsigned long long)__double_as_longlong(shifted) - 0x41B8000000000000ull,
                                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                    }
                                }
                            } else {
                                if (ok_a) {
                                    double *tp = ls + base;
#pragma unroll
                                    for (int i = 0; i < VEC; i++)
                                        __hip_atomic_fetch_add(tp + i, (double)val[u][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                }
                            }
                            if (plain) {
                                if (ok_a && ch0[u] == 0) atomicAdd(&lc[lab_a[u]], 1u);
                            } else if (ok_a) {
#pragma unroll
                                for (int i = 0; i < VEC; i++) {
                                    TableAdd<T>::add(ls, (size_t)lab_a[u] * cs + ch0[u] + i, val[u][i], sums, (size_t)lab_a[u] * c + ch0[u] + i);
                                    if (ch0[u] + i == 0) atomicAdd(&lc[lab_a[u]], 1u);
                                }
                            }
                        }
                        continue;
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        bool plain = false;
                        if constexpr (TableAdd<T>::kFixed) {
                            // binary16, no Inf / NaN among the eight (the usual vector): branch-free.  v + 1.5 * 2^28 is
                            // exact and lies in [2^28, 2^29), where one ulp is 2^-24: its bit pattern minus the
                            // constant's IS v * 2^24 in two's complement (the constant's low word is zero: one
                            // subtraction on the high word).  Elements without a valid label go to the spare slots.
                            // (an exponent field of all ones <=> magnitude >= 0x7c00 <=> magnitude + 0x0400 reaches bit 15;
                            // both halves of a word at once, no carry between them)
                            unsigned raw[4], reach = 0u;
                            __builtin_memcpy(raw, val[u], 16);
#pragma unroll
                            for (int d = 0; d < 4; d++) reach |= (raw[d] & 0x7fff7fffu) + 0x04000400u;
                            plain = (reach & 0x80008000u) == 0u;
                            if (plain) {
                                const bool ok_a = (unsigned)lab_a[u] < (unsigned)k, ok_b = (unsigned)lab_b[u] < (unsigned)k;
                                const int base_a = ok_a ? lab_a[u] * cs + ch0[u] : k * cs;
                                const int base_b = ok_b ? lab_b[u] * cs + ch0[u] - c : k * cs;
                                unsigned long long *table = reinterpret_cast<unsigned long long *>(ls);
#pragma unroll
                                for (int i = 0; i < VEC; i++) {
                                    const double shifted = (double)val[u][i] + 0x1.8p+28;
                                    const unsigned long long q =
                                        (unsigned long long)__double_as_longlong(shifted) - 0x41B8000000000000ull;
                                    const int slot = (ch0[u] + i >= c ? base_b : base_a) + i;
                                    if (q) __hip_atomic_fetch_add(table + slot, q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                }
                                if (ok_a && ch0[u] == 0) atomicAdd(&lc[lab_a[u]], 1u);        // the vector starts a row,
                                if (ok_b && ch0[u] + VEC > c) atomicAdd(&lc[lab_b[u]], 1u);   // or the next row starts inside it
                            }
                        }
                        if (!plain) {
#pragma unroll
                            for (int i = 0; i < VEC; i++) {
                                const bool wrapped = ch0[u] + i >= c;
                                const int lb = wrapped ? lab_b[u] : lab_a[u];
                                const int ch = ch0[u] + i - (wrapped ? c : 0);
                                if (lb >= 0 && lb < k) {
                                    TableAdd<T>::add(ls, (size_t)lb * cs + ch, val[u][i], sums, (size_t)lb * c + ch);
                                    if (ch == 0) atomicAdd(&lc[lb], 1u);
                                }
                            }
                        }
                    }
                }
            } else
            for (int64_t v0 = tid; v0 < nvec; v0 += 4 * NT) {
                T val[4][VEC];
                int lab[4][VEC], chn[4][VEC];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int64_t v = v0 + u * NT;
                    const bool ok = v < nvec;
                    const u4 raw = ok ? *reinterpret_cast<const u4 *>(xb + VEC * v) : u4{0u, 0u, 0u, 0u};
                    __builtin_memcpy(val[u], &raw, 16);
                    int64_t row = vrow;
                    int ch = vch;
                    vrow += drow;
                    vch += dch;
                    if (vch >= c) {
                        vch -= c;
                        vrow++;
                    }
                    int lb = ok ? labels[r0 + row] - 1 : -1;
#pragma unroll
                    for (int i = 0; i < VEC; i++) {
                        chn[u][i] = ch;
                        lab[u][i] = lb;
                        if (++ch == c) {
                            ch = 0;
                            row++;
                            lb = (ok && r0 + row < r1) ? labels[r0 + row] - 1 : -1;
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; u++)
#pragma unroll
                    for (int i = 0; i < VEC; i++) {
                        const int lb = lab[u][i];
                        if (lb >= 0 && lb < k) {
                            TableAdd<T>::add(ls, (size_t)lb * cs + chn[u][i], val[u][i], sums, (size_t)lb * c + chn[u][i]);
                            if (chn[u][i] == 0) atomicAdd(&lc[lb], 1u);
                        }
                    }
            }
            // the (total % VEC) trailing elements of the range
            for (int64_t e = nvec * VEC + tid; e < total; e += NT) {
                const int64_t row = e / c;
                const int ch = (int)(e - row * c), lb = labels[r0 + row] - 1;
                if (lb >= 0 && lb < k) {
                    TableAdd<T>::add(ls, (size_t)lb * cs + ch, xb[e], sums, (size_t)lb * c + ch);
                    if (ch == 0) atomicAdd(&lc[lb], 1u);
                }
            }
            done = true;
        }
    }
    if (r0 < r1 && !done) {
        // element e of the range <-> (row r0 + e / c, channel e % c); advance by 256 per element,
        // four elements in flight per thread (loads issued before the dependent atomics)
        const int64_t total = (r1 - r0) * c;
        int64_t row = r0 + tid / c;
        int ch = tid % c;
        const int drow = NT / c, dch = NT % c;
        for (int64_t e = tid; e < total; e += 4 * NT) {
            int cc[4], lab[4];
            T v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                cc[u] = ch;
                const bool ok = e + NT * u < total;
                lab[u] = ok ? labels[row] - 1 : -1;
                v[u] = ok ? x[row * ldx + ch] : (T)0;
                row += drow;
                ch += dch;
                if (ch >= c) {
                    ch -= c;
                    row += 1;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if constexpr (sizeof(T) == 8) v[u] = pxsom_bmu::qround(v[u], qmagic);   // binary64 rows of a reproducible run
                if (lab[u] >= 0 && lab[u] < k) {
                    if (use_lds) {
                        TableAdd<T>::add(ls, (size_t)lab[u] * cs + cc[u], v[u], sums, (size_t)lab[u] * c + cc[u]);
                        if (cc[u] == 0) atomicAdd(&lc[lab[u]], 1u);
                    } else {
                        __hip_atomic_fetch_add(&sums[(size_t)lab[u] * c + cc[u]], (double)v[u], __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
                        if (cc[u] == 0) {
                            if constexpr (COUNT_F64)
                                __hip_atomic_fetch_add(reinterpret_cast<double *>(counts) + lab[u], 1.0, __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_AGENT);
                            else
                                atomicAdd(&counts[lab[u]], 1ull);
                        }
                    }
                }
            }
        }
    }
    if (use_lds) {
        __syncthreads();
        int node = tid / c, j = tid - node * c;   // element e <-> (node, channel), advanced without a division per element
        const int dnode = NT / c, dj = NT % c;
        for (int e = tid; e < k * c; e += NT) {
            const double v = TableAdd<T>::value(ls, (size_t)node * cs + j);
            if (v != 0.0) __hip_atomic_fetch_add(&sums[e], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            node += dnode;
            j += dj;
            if (j >= c) {
                j -= c;
                node++;
            }
        }
        for (int e = tid; e < k; e += NT)
            if (lc[e]) {
                if constexpr (COUNT_F64)
                    __hip_atomic_fetch_add(reinterpret_cast<double *>(counts) + e, (double)lc[e], __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                else
                    atomicAdd(&counts[e], (unsigned long long)lc[e]);
            }
    }
}

template <typename T, int CMAX, int MAXT>
int launch_online(const T *x, int64_t n, int c, int64_t ldx, double *w, int xdim, int ydim, int rlen, double a0,
                  double a1, double r0, double r1, const int64_t *order, int flags, hipStream_t st)
{
    const int K = xdim * ydim;
    const int bd = ((K + 63) / 64) * 64;
    const int nwv = bd / 64;
    const int cs = CMAX > 0 ? CMAX : c;
    // LDS besides the row ring: the codebook (CMAX == 0, when it fits) + per-wave exchange + learning rates
    auto plan = [&](bool codebook_in_lds, int *chunk_out) -> size_t {
        const size_t fixed = (codebook_in_lds ? (size_t)c * K * 8 : 0) + (size_t)2 * nwv * 8 + (size_t)2 * nwv * 4 +
                             (size_t)nwv * 8 + 2 * 64 * 8 + 64;
        int chunk = 64;
        while (chunk > 8 && fixed + (size_t)2 * chunk * cs * 8 > 150 * 1024) chunk >>= 1;
        while ((chunk * c + bd - 1) / bd > 16) chunk >>= 1;  // gather registers per thread
        *chunk_out = chunk;
        return fixed + (size_t)2 * chunk * cs * 8;
    };
    int chunk = 0;
    size_t lds = plan(CMAX == 0, &chunk);
    bool in_place = false;
    if constexpr (CMAX == 0) {
        if (chunk < 1 || lds > 160 * 1024) {   // the codebook does not fit beside the ring: train it where it lies
            in_place = true;
            lds = plan(false, &chunk);
        }
    }
    if (chunk < 1 || lds > 160 * 1024)
        return pxsom::fail(PXSOM_ERR_UNSUPPORTED, "pxsom_train_online: %d nodes x %d channels: no LDS for the row ring", K, c);
    auto launch = [&](auto kern) -> int {
        PXSOM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(1), dim3(bd), lds, st, x, n, c, ldx, w, xdim, ydim, rlen, a0, a1, r0, r1,
                           order, chunk, flags);
        PXSOM_LAUNCH_CHECK("som_online_kernel");
        return PXSOM_OK;
    };
    if constexpr (CMAX == 0) {
        if (in_place) return launch(som_online_kernel<T, CMAX, MAXT, true>);
    }
    return launch(som_online_kernel<T, CMAX, MAXT, false>);
}

template <typename T, int CH, int L>
int launch_online_split(const T *x, int64_t n, int c, int64_t ldx, double *w, int xdim, int ydim, int rlen,
                        double a0, double a1, double r0, double r1, const int64_t *order, int flags, hipStream_t st)
{
    const int K = xdim * ydim;
    const int bd = ((K * L + 63) / 64) * 64;
    int chunk = 64;
    while ((chunk * c + bd - 1) / bd > 16) chunk >>= 1;  // gather registers per thread
    const size_t lds = (size_t)2 * chunk * CH * L * 8 + (3 * 128 + 8) * 8 + (size_t)chunk * 8 +
                       (size_t)2 * chunk * 8 + (size_t)(CH * L + 2) * 8;
    auto kern = som_online_split_kernel<T, CH, L>;
    if (lds > 48 * 1024)
        PXSOM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(1), dim3(bd), lds, st, x, n, c, ldx, w, xdim, ydim, rlen, a0, a1, r0, r1,
                       order, chunk, flags);
    PXSOM_LAUNCH_CHECK("som_online_split_kernel");
    return PXSOM_OK;
}

template <typename T>
int train_online_typed(const T *x, int64_t n, int c, int64_t ldx, double *w, int xdim, int ydim, int rlen,
                       double a0, double a1, double r0, double r1, const int64_t *order, int flags, hipStream_t st)
{
#define PXSOM_ONLINE(CM, MT) \
    return launch_online<T, CM, MT>(x, n, c, ldx, w, xdim, ydim, rlen, a0, a1, r0, r1, order, flags, st)
#define PXSOM_ONLINE_SPLIT(CH, L) \
    return launch_online_split<T, CH, L>(x, n, c, ldx, w, xdim, ydim, rlen, a0, a1, r0, r1, order, flags, st)
    // small maps: several lanes per node (fewer binary64 instructions per wave per step)
    if (xdim * ydim <= 64 && c <= 40) {
        if (c <= 16) PXSOM_ONLINE_SPLIT(4, 4);
        if (c <= 24) PXSOM_ONLINE_SPLIT(6, 4);
        PXSOM_ONLINE_SPLIT(10, 4);
    }
    if (xdim * ydim <= 128 && c <= 40) {
        if (c <= 8) PXSOM_ONLINE_SPLIT(4, 2);
        if (c <= 16) PXSOM_ONLINE_SPLIT(8, 2);
        if (c <= 24) PXSOM_ONLINE_SPLIT(12, 2);
        PXSOM_ONLINE_SPLIT(20, 2);
    }
    // wide rows (cell SOM: ~100 cluster-count features) on maps up to 128 nodes: 4 lanes per node in a
    // 512-thread workgroup (two waves per SIMD)
    if (xdim * ydim <= 128 && c <= 104) {
        if (c <= 64) PXSOM_ONLINE_SPLIT(16, 4);
        if (c <= 80) PXSOM_ONLINE_SPLIT(20, 4);
        PXSOM_ONLINE_SPLIT(26, 4);
    }
#undef PXSOM_ONLINE_SPLIT
    // <= 256 nodes: 4 waves at most, the whole register file is available per thread
    if (xdim * ydim <= 256) {
        if (c <= 8) PXSOM_ONLINE(8, 256);
        if (c <= 16) PXSOM_ONLINE(16, 256);
        if (c <= 24) PXSOM_ONLINE(24, 256);
        if (c <= 40) PXSOM_ONLINE(40, 256);
        if (c <= 64) PXSOM_ONLINE(64, 256);
        if (c <= 104) PXSOM_ONLINE(104, 256);   // cell SOM: ~100 cluster-count features (one wave per SIMD: 512 registers)
        PXSOM_ONLINE(0, 256);
    }
    // <= 512 nodes (config 5's 20 x 20 map): two waves per SIMD, 256 registers per thread
    if (xdim * ydim <= 512) {
        if (c <= 8) PXSOM_ONLINE(8, 512);
        if (c <= 16) PXSOM_ONLINE(16, 512);
        if (c <= 24) PXSOM_ONLINE(24, 512);
        if (c <= 40) PXSOM_ONLINE(40, 512);
    }
    PXSOM_ONLINE(0, 1024);  // more nodes or wider rows: 128 VGPRs per thread, codebook stays in LDS
#undef PXSOM_ONLINE
}

// ------------------------------------------------------------------------------------------------
// per-cluster sums, wave-private tables: LDS binary64 atomics retire about one lane per clock per CU, which
// holds the atomic kernel above at ~2.5 TB/s.  Here every wave owns a [k + 1, c] table in LDS and updates
// it with plain read / add / write: RPI = 64 / c rows per instruction, lane <-> (row slot, channel), so the
// lanes of one instruction touch distinct words unless two of its rows carry the same label.  Two groups
// (2 * RPI rows, a "unit") are applied together -- both reads, both adds, both writes -- when no label
// repeats inside the unit; the test is one ballot per 64 labels (each lane compares its row's label with
// the others of its unit), read per unit as a few bits of a scalar mask.  A unit with a repeat is applied
// row by row.  LDS executes a wave's accesses in order, so the writes of one unit precede the reads of
// the next without any wait.  Rows outside [ra, rb) and labels outside 1..k go to the spare row k; lanes
// past RPI * c to spare words behind the table.
// Loads: one tile (RPI * U rows) ahead, each value register re-issued for the next tile right after its
// use (U dword buffer loads in flight per lane, scalar group offset); labels two tiles ahead.  Measured
// (10x10 x 22, 4.2 M float32 rows): 110 us against 160 us for the atomic kernel; the load stream alone
// runs at ~4.7 TB/s in this one-dword-per-lane shape (82 us), the table updates add the rest.
// ------------------------------------------------------------------------------------------------
template <typename T, int RPI, bool COUNT_F64>
__global__ __launch_bounds__(256) void cluster_sums_private_kernel(const T *__restrict__ x, int64_t n, int c,
                                                                   int64_t ldx,
                                                                   const int32_t *__restrict__ labels, int k,
                                                                   double *sums, unsigned long long *counts,
                                                                   int64_t rows_per_wave)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int U = sizeof(T) == 8 ? 16 : 32;  // groups per tile == loads in flight per lane (56: no gain)
    constexpr int TR = RPI * U;                  // rows per tile (<= 128)
    constexpr int SG = 2 * RPI;                  // rows of two groups: the unit whose labels are compared
    constexpr int RL = 64 / SG * SG;             // rows per label register (whole units)
    constexpr int NL = (TR + RL - 1) / RL;       // label registers per tile
    const int tid = threadIdx.x, bd = blockDim.x, lane = tid & 63, nwv = bd >> 6;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // keeps the row arithmetic on the scalar unit
    const int tstride = (k + 1) * c + 64;        // doubles per wave table
    double *all = reinterpret_cast<double *>(smem_raw);
    double *tbl = all + (size_t)wv * tstride;
    unsigned *cnt = reinterpret_cast<unsigned *>(all + (size_t)nwv * tstride);  // [k], shared by the waves
    for (int e = tid; e < nwv * tstride; e += bd) all[e] = 0.0;
    for (int e = tid; e < k; e += bd) cnt[e] = 0u;
    __syncthreads();

    const int slot = lane / c, ch = lane - slot * c;
    const bool active = slot < RPI;
    const unsigned lane_off = active ? (unsigned)((slot * ldx + ch) * (int64_t)sizeof(T)) : 0u;  // bytes
    const int spare = (k + 1) * c + lane;  // lanes past RPI * c: a word of their own behind the table
    const int64_t gw = (int64_t)blockIdx.x * nwv + wv;
    const int64_t ra = gw * rows_per_wave;
    const int64_t rb = ra + rows_per_wave < n ? ra + rows_per_wave : n;
    if (ra < rb) {  // wave-uniform
        // Whole groups end at row `lim` (relative to ra); the (n - ra) % RPI rows behind it (last wave only)
        // are added one by one.  Every load is unconditional with a clamped, wave-uniform row (groups past
        // the end re-read the rows at `safe`): a load behind a branch costs an s_waitcnt vmcnt(0) per group.
        const int span = (int)(rb - ra), lim = span - span % RPI;
        const int last = (int)(n - 1 - ra);  // last row of the matrix, relative
        // buffer loads: wave-uniform descriptor + 32-bit lane offset + scalar group offset (no 64-bit
        // address arithmetic per load).  The launcher keeps a wave's byte range below 2^31.
        const unsigned gstep = (unsigned)(RPI * ldx * (int64_t)sizeof(T));  // bytes from one group to the next
        const unsigned safe = (unsigned)((ra + RPI <= n ? 0 : n - RPI - ra) * ldx * (int64_t)sizeof(T));
        const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<T *>(x + ra * ldx), 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t lres = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<int32_t *>(labels + ra), 0, 0x7fffffff, 0x00020000);
        auto load_labels = [&](int rel0, int(&lv)[NL]) {
#pragma unroll
            for (int i = 0; i < NL; i++) {
                const int r = rel0 + i * RL + lane;
                const int lb = __builtin_amdgcn_raw_buffer_load_b32(lres, (r < last ? r : last) * 4, 0, 0) - 1;
                // '&', not '&&': a short-circuit lets the compiler sink the load into a branch
                const bool ok = (r < lim) & (lane < RL) & (i * RL + lane < TR) & ((unsigned)lb < (unsigned)k);
                lv[i] = ok ? lb : k;
            }
        };
        auto load_val = [&](int rel, unsigned off) -> T {  // group at relative row rel, byte offset off
            const unsigned so = rel + RPI <= lim ? off : safe;
            if constexpr (sizeof(T) == 8) {
                typedef unsigned u2 __attribute__((ext_vector_type(2)));
                const u2 raw = __builtin_amdgcn_raw_buffer_load_b64(xres, lane_off, so, 0);
                return __builtin_bit_cast(T, raw);
            } else if constexpr (sizeof(T) == 4) {
                return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b32(xres, lane_off, so, 0));
            } else {
                return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b16(xres, lane_off, so, 0));
            }
        };
        // labels shared inside a unit of 2 groups: such a unit is applied row by row
        auto clash_mask = [&](int lab) -> unsigned long long {
            const int base = lane / SG * SG, pos = lane - base;
            bool cl = false;
#pragma unroll
            for (int d = 1; d < SG; d++) {
                const int p = pos + d < SG ? pos + d : pos + d - SG;
                cl = cl | (__shfl(lab, base + p) == lab);  // every lane takes part in every exchange
            }
            return __ballot(cl && lab != k && lane < RL);
        };
        T val[U];
        // labels travel two tiles ahead and are requested BEFORE the tile's value loads, so waiting for
        // them never drains the value loads behind them (vmcnt counts in issue order)
        int lv_cur[NL], lv_nxt[NL], lv_far[NL];
        load_labels(0, lv_cur);
        load_labels(TR, lv_nxt);
        {
            unsigned off = 0;
#pragma unroll
            for (int g = 0; g < U; g++, off += gstep) val[g] = load_val(g * RPI, off);
        }
        // byte address of this lane's word in the table row of a label: tbl + (label * c + ch) * 8
        char *const lane_word = reinterpret_cast<char *>(tbl) + (active ? ch : spare) * 8;
        unsigned tile_off = TR * gstep / RPI;  // byte offset of the next tile
        for (int rel0 = 0; rel0 < lim; rel0 += TR, tile_off += TR * gstep / RPI) {
            load_labels(rel0 + 2 * TR, lv_far);
            unsigned long long cm[NL];
            int row_bytes[NL];  // label * c * 8 of the rows this lane holds
#pragma unroll
            for (int i = 0; i < NL; i++) {
                if (lv_cur[i] < k) atomicAdd(&cnt[lv_cur[i]], 1u);
                cm[i] = clash_mask(lv_cur[i]);
                row_bytes[i] = (int)__umul24(lv_cur[i], c * 8);
            }
            // every group's table address for this lane, one exchange each, all issued before the first use
            int word[U];
#pragma unroll
            for (int g = 0; g < U; g++) {
                const int r = g * RPI;
                const int rb8 = __shfl(row_bytes[r / RL], r % RL + slot);
                word[g] = active ? rb8 : 0;
            }
            unsigned off = tile_off;
#pragma unroll
            for (int g = 0; g < U; g += 2) {
                double *const w0 = reinterpret_cast<double *>(lane_word + word[g]);
                double *const w1 = reinterpret_cast<double *>(lane_word + word[g + 1]);
                const double v0 = (double)val[g], v1 = (double)val[g + 1];
                val[g] = load_val(rel0 + TR + g * RPI, off);  // these registers' loads for the next tile
                val[g + 1] = load_val(rel0 + TR + (g + 1) * RPI, off + gstep);
                off += 2 * gstep;
                if (!((cm[g * RPI / RL] >> (g * RPI % RL)) & ((1ull << SG) - 1))) {  // no row of the unit flagged
                    const double a = *w0, b = *w1;
                    *w0 = a + v0;
                    *w1 = b + v1;
                } else {
                    // one row at a time.  The fences keep the predicated updates apart: to the compiler they are
                    // mutually exclusive branches of one thread, which it may fold into a single update
#pragma unroll
                    for (int s = 0; s < RPI; s++) {
                        if (slot == s) *w0 += v0;
                        __builtin_amdgcn_wave_barrier();
                        asm volatile("" ::: "memory");
                    }
#pragma unroll
                    for (int s = 0; s < RPI; s++) {
                        if (slot == s) *w1 += v1;
                        __builtin_amdgcn_wave_barrier();
                        asm volatile("" ::: "memory");
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < NL; i++) {
                lv_cur[i] = lv_nxt[i];
                lv_nxt[i] = lv_far[i];
            }
        }
        for (int64_t r = ra + lim; r < rb; r++) {  // fewer than RPI rows
            const int lb = labels[r] - 1;
            if ((unsigned)lb < (unsigned)k) {
                if (lane < c) tbl[lb * c + lane] += (double)x[r * ldx + lane];
                if (lane == 0) atomicAdd(&cnt[lb], 1u);
            }
        }
    }
    __syncthreads();
    for (int e = tid; e < k * c; e += bd) {
        double v = 0.0;
        for (int w = 0; w < nwv; w++) v += all[(size_t)w * tstride + e];
        if (v != 0.0) __hip_atomic_fetch_add(&sums[e], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int e = tid; e < k; e += bd)
        if (cnt[e]) {
            if constexpr (COUNT_F64)
                __hip_atomic_fetch_add(reinterpret_cast<double *>(counts) + e, (double)cnt[e], __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            else
                atomicAdd(&counts[e], (unsigned long long)cnt[e]);
        }
}

// wave-private form when c <= 64 and at least one wave's table fits the CU; *handled says whether it ran
template <typename T, int RPI, bool COUNT_F64>
int launch_sums_private(const T *x, int64_t n, int c, int64_t ldx, const int32_t *labels, int k, double *sums,
                        int64_t *counts, hipStream_t st, int nwv, int blocks_per_cu)
{
    const size_t tbytes = ((size_t)(k + 1) * c + 64) * 8;
    const size_t lds = tbytes * nwv + (size_t)k * 4;
    constexpr int TR = RPI * (sizeof(T) == 8 ? 16 : 32);
    const int64_t max_waves = (int64_t)pxsom::device_cu_count() * blocks_per_cu * nwv;
    // every wave gets whole tiles, and enough of them to pay for its share of the final merge
    int64_t rows_per_wave = (n + max_waves - 1) / max_waves;
    if (rows_per_wave < 8 * TR) rows_per_wave = 8 * TR;
    rows_per_wave = (rows_per_wave + TR - 1) / TR * TR;
    const int64_t waves = (n + rows_per_wave - 1) / rows_per_wave;
    const int64_t grid = (waves + nwv - 1) / nwv;
    auto kern = cluster_sums_private_kernel<T, RPI, COUNT_F64>;
    if (lds > 48 * 1024)
        PXSOM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * nwv), lds, st, x, n, c, ldx, labels, k, sums,
                       reinterpret_cast<unsigned long long *>(counts), rows_per_wave);
    PXSOM_LAUNCH_CHECK("cluster_sums_private_kernel");
    return PXSOM_OK;
}

// Wave-private tables (cluster_sums_private_kernel, the pairs kernel): how many waves of a workgroup get one (0: the shape has no
// such route) and how many workgroups a CU holds.
inline int sums_private_waves(int c, int k, int *per_cu_out = nullptr)
{
    int nwv = 0, per_cu = 1;
    if (c >= 13 && c <= 64) {
        const size_t tbytes = ((size_t)(k + 1) * c + 64) * 8, budget = 160 * 1024 - 1024;
        if (8 * tbytes + (size_t)k * 8 <= budget) nwv = 4, per_cu = 2;
        else if (4 * tbytes + (size_t)k * 4 <= budget) nwv = 4;
        else if (2 * tbytes + (size_t)k * 4 <= budget) nwv = 2;
    }
    if (per_cu_out) *per_cu_out = per_cu;
    return nwv;
}

// The shapes whose sums kernel reads a scheduled step's rows where they lie (pxsom::RowView): cluster_sums_kernel's flat route
// with rows of whole 16-byte vectors -- binary32 / binary16 rows, contiguous in the caller's matrix, table in LDS -- and no
// wave-private route for the shape (those kernels address flat ranges; large inputs of 13 - 64 channels go there when their tables
// fit: config 5's 400 x 40 table does not).
template <typename T>
bool sums_take_views(const T *x, int c, int64_t ldx, int k)
{
    if (sizeof(T) > 4) return false;
    const int vec = 16 / (int)sizeof(T);
    const size_t lds_odd = ((size_t)k * (c | 1) + kSumsSpare) * 8 + (size_t)k * 4;
    return sums_private_waves(c, k) == 0 && c >= vec && c % vec == 0 && ldx == c && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
           std::min(lds_odd, ((size_t)k * c + kSumsSpare) * 8 + (size_t)k * 4) <= 150 * 1024;
}

// qmagic != 0 (binary64 rows of a reproducible training run, include/pxsom.h): values are rounded to the run's quantum
// as they are added; only the atomic kernel knows how.
template <typename T, bool COUNT_F64 = false>
int cluster_sums_typed(const T *x, int64_t n, int c, int64_t ldx, const int32_t *labels, int k, double *sums,
                       int64_t *counts, hipStream_t st, double qmagic = 0.0)
{
    if (sizeof(T) != 8) qmagic = 0.0;
    if (pxsom::row_view_active() && !sums_take_views<T>(x, c, ldx, k))
        return pxsom::fail(PXSOM_ERR_UNSUPPORTED, "cluster sums: a row view on a shape whose kernel does not take views");
    // wave-private tables (see cluster_sums_private_kernel): 13 <= c <= 64 (RPI = 64 / c <= 4 rows per
    // instruction, fewer idle lanes than channels), at least two tables per CU, an input big enough to
    // fill them, and a wave's byte range addressable by the 32-bit buffer offsets.  Measured against the
    // atomic kernel below on 2-4 M rows: 1.4-1.9x faster there, slower outside (c <= 8, one table per CU).
    if (c >= 13 && c <= 64 && n >= 32768 && qmagic == 0.0) {
        int per_cu = 1;
        const int nwv = sums_private_waves(c, k, &per_cu);
        const int64_t waves = (int64_t)pxsom::device_cu_count() * per_cu * (nwv ? nwv : 1);
        const bool addressable = ((n + waves - 1) / waves + 1024) * ldx * (int64_t)sizeof(T) < (1ll << 31);
        if (nwv && addressable) {
            // two channels per lane where the rows allow pair loads (pxsom_sums.hip): 2.5x fewer instructions per row
            if (pxsom::launch_sums_pairs<T>(x, n, c, ldx, labels, k, sums, counts, COUNT_F64, st, nwv, per_cu)) {
                PXSOM_LAUNCH_CHECK("cluster_sums_pairs_kernel");
                return PXSOM_OK;
            }
            switch (64 / c) {
                case 1: return launch_sums_private<T, 1, COUNT_F64>(x, n, c, ldx, labels, k, sums, counts, st, nwv, per_cu);
                case 2: return launch_sums_private<T, 2, COUNT_F64>(x, n, c, ldx, labels, k, sums, counts, st, nwv, per_cu);
                case 3: return launch_sums_private<T, 3, COUNT_F64>(x, n, c, ldx, labels, k, sums, counts, st, nwv, per_cu);
                default: return launch_sums_private<T, 4, COUNT_F64>(x, n, c, ldx, labels, k, sums, counts, st, nwv, per_cu);
            }
        }
    }
    // table row stride: odd when the padded table still fits (bank spread of the LDS atomics), else c
    const size_t lds_odd = ((size_t)k * (c | 1) + kSumsSpare) * 8 + (size_t)k * 4;
    const int cs = lds_odd <= 150 * 1024 ? (c | 1) : c;
    const size_t lds = ((size_t)k * cs + kSumsSpare) * 8 + (size_t)k * 4;
    const int use_lds = lds <= 150 * 1024;
    const int cus = pxsom::device_cu_count();
    // small inputs are latency-bound per workgroup, so they are spread wide: 64 rows per workgroup (measured on
    // config 4's 15.6 K-row training steps, ms per 64-step pass: 32 rows 6.09, 64 rows 5.86, 128 rows 6.08, 512 rows 9.5)
    constexpr int rows_per_wg = 64;
    const int wg_per_cu = (int)std::max<size_t>(1, std::min<size_t>(4, (size_t)(158 * 1024) / std::max<size_t>(lds, 1)));
    int64_t grid = std::min<int64_t>((n + rows_per_wg - 1) / rows_per_wg, (int64_t)cus * wg_per_cu);
    if (grid < 1) grid = 1;
    // (multiples of 16 rows keep every workgroup's range 16-byte aligned for the vector loads)
    const int64_t rows_per_block = ((n + grid - 1) / grid + 15) / 16 * 16;
    // one table per CU (K = 400 x C = 40: 128 KB): 16 waves share it, so that enough loads and LDS atomics are in
    // flight; small tables keep 4-wave workgroups (several per CU)
    const bool wide = use_lds && lds > 64 * 1024 && n >= 65536;
    auto kern = wide ? cluster_sums_kernel<T, COUNT_F64, 1024> : cluster_sums_kernel<T, COUNT_F64, 256>;
    if (use_lds && lds > 48 * 1024)
        PXSOM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(wide ? 1024 : 256), use_lds ? lds : 0, st, x, n, c, ldx, labels, k,
                       sums, reinterpret_cast<unsigned long long *>(counts), rows_per_block, use_lds, qmagic, cs, pxsom::current_row_view());
    PXSOM_LAUNCH_CHECK("cluster_sums_kernel");
    return PXSOM_OK;
}

int check_matrix(const char *fn, const void *x, int64_t n, int c, int64_t ldx, int dtype)
{
    if (n < 0) return pxsom::fail(PXSOM_ERR_INVALID_ARG, "%s: n=%lld < 0", fn, (long long)n);
    if (c < 1 || c > PXSOM_MAX_CHANNELS)
        return pxsom::fail(PXSOM_ERR_UNSUPPORTED, "%s: c=%d outside [1, %d]", fn, c, PXSOM_MAX_CHANNELS);
    if (ldx < c) return pxsom::fail(PXSOM_ERR_INVALID_ARG, "%s: ldx=%lld < c=%d", fn, (long long)ldx, c);
    if (!pxsom::dtype_ok(dtype)) return pxsom::fail(PXSOM_ERR_UNSUPPORTED, "%s: dtype %d", fn, dtype);
    if (n > 0 && !x) return pxsom::fail(PXSOM_ERR_INVALID_ARG, "%s: null matrix", fn);
    return PXSOM_OK;
}

}  // namespace

PXSOM_EXPORT int pxsom_train_online_ex(const void *x_dev, int64_t n, int c, int64_t ldx, int dtype, double *w_dev,
                                       int xdim, int ydim, int rlen, double a0, double a1, double r0, double r1,
                                       const int64_t *order_dev, int flags, void *stream)
{
    int rc = check_matrix("pxsom_train_online", x_dev, n, c, ldx, dtype);
    if (rc) return rc;
    if (xdim < 1 || ydim < 1 || (int64_t)xdim * ydim > PXSOM_MAX_NODES)
        return pxsom::fail(PXSOM_ERR_UNSUPPORTED, "pxsom_train_online: grid %dx%d outside [1, %d] nodes", xdim,
                           ydim, PXSOM_MAX_NODES);
    if (rlen < 0 || !w_dev) return pxsom::fail(PXSOM_ERR_INVALID_ARG, "pxsom_train_online: bad rlen / null codebook");
    if (flags & ~PXSOM_ONLINE_INT_ABS) return pxsom::fail(PXSOM_ERR_INVALID_ARG, "pxsom_train_online: unknown flags %d", flags);
    if (n == 0 || rlen == 0) return PXSOM_OK;
    if (!order_dev) return pxsom::fail(PXSOM_ERR_INVALID_ARG, "pxsom_train_online: null order");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    PXSOM_DISPATCH_DTYPE(dtype, x_dev, xp,
                         train_online_typed<T>(xp, n, c, ldx, w_dev, xdim, ydim, rlen, a0, a1, r0, r1, order_dev, flags, st));
}

PXSOM_EXPORT int pxsom_train_online(const void *x_dev, int64_t n, int c, int64_t ldx, int dtype, double *w_dev,
                                    int xdim, int ydim, int rlen, double a0, double a1, double r0, double r1,
                                    const int64_t *order_dev, void *stream)
{
    return pxsom_train_online_ex(x_dev, n, c, ldx, dtype, w_dev, xdim, ydim, rlen, a0, a1, r0, r1, order_dev, 0, stream);
}

PXSOM_EXPORT int pxsom_cluster_sums(const void *x_dev, int64_t n, int c, int64_t ldx, int dtype,
                                    const int32_t *labels_dev, int k, double *sums_dev, int64_t *counts_dev,
                                    void *stream)
{
    int rc = check_matrix("pxsom_cluster_sums", x_dev, n, c, ldx, dtype);
    if (rc) return rc;
    if (k < 1 || k > PXSOM_MAX_NODES)
        return pxsom::fail(PXSOM_ERR_UNSUPPORTED, "pxsom_cluster_sums: k=%d outside [1, %d]", k, PXSOM_MAX_NODES);
    if (!sums_dev || !counts_dev || (n > 0 && !labels_dev))
        return pxsom::fail(PXSOM_ERR_INVALID_ARG, "pxsom_cluster_sums: null pointer");
    if (n == 0) return PXSOM_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    PXSOM_DISPATCH_DTYPE(dtype, x_dev, xp, cluster_sums_typed<T>(xp, n, c, ldx, labels_dev, k, sums_dev, counts_dev, st));
}

// cell x pixel-cluster counts (create_c2pc_data's groupby + pivot): plain global int64 atomics.  The bins
// of one cell are nb consecutive words and neighbouring pixels mostly belong to the same cell, so the
// atomics of a wave land in a few cache lines; the kernel is bound by reading the two label vectors.
__global__ __launch_bounds__(256) void pair_histogram_kernel(const int32_t *__restrict__ a,
                                                             const int32_t *__restrict__ b, int64_t n,
                                                             int64_t na, int nb, unsigned long long *hist)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int32_t ai = a[i], bi = b[i];
        if (ai >= 0 && ai < na && bi >= 0 && bi < nb) atomicAdd(&hist[(int64_t)ai * nb + bi], 1ull);
    }
}

PXSOM_EXPORT int pxsom_pair_histogram(const int32_t *a_dev, const int32_t *b_dev, int64_t n, int64_t na, int nb,
                                      int64_t *hist_dev, void *stream)
{
    if (n < 0 || na < 1 || nb < 1) return pxsom::fail(PXSOM_ERR_INVALID_ARG, "pxsom_pair_histogram: bad sizes");
    if (!hist_dev || (n > 0 && (!a_dev || !b_dev)))
        return pxsom::fail(PXSOM_ERR_INVALID_ARG, "pxsom_pair_histogram: null pointer");
    if (n == 0) return PXSOM_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t grid = std::min<int64_t>((n + 255) / 256, (int64_t)pxsom::device_cu_count() * 16);
    hipLaunchKernelGGL(pair_histogram_kernel, dim3((unsigned)grid), dim3(256), 0, st, a_dev, b_dev, n, na, nb,
                       reinterpret_cast<unsigned long long *>(hist_dev));
    PXSOM_LAUNCH_CHECK("pair_histogram_kernel");
    return PXSOM_OK;
}

PXSOM_EXPORT int pxsom_batch_update(double *w_dev, int xdim, int ydim, int c, const double *sums_dev,
                                    const double *counts_dev, double thr, double alpha, void *stream)
{
    if (xdim < 1 || ydim < 1 || (int64_t)xdim * ydim > PXSOM_MAX_NODES || c < 1 || c > PXSOM_MAX_CHANNELS)
        return pxsom::fail(PXSOM_ERR_UNSUPPORTED, "pxsom_batch_update: shape %dx%d x %d", xdim, ydim, c);
    if (!w_dev || !sums_dev || !counts_dev) return pxsom::fail(PXSOM_ERR_INVALID_ARG, "pxsom_batch_update: null pointer");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int K = xdim * ydim;
    // statistics of the widest window (the whole grid) staged in LDS when they fit
    const size_t stage_bytes = (size_t)K * (c + 1) * sizeof(double);
    const int stage = stage_bytes <= 60 * 1024;
    hipLaunchKernelGGL(batch_update_kernel, dim3(K), dim3(256), stage ? stage_bytes : 0, st, w_dev, xdim, ydim, c,
                       sums_dev, counts_dev, thr, 1.0 - alpha, pxsom_bmu::batch_gain_saturation(1.0 - alpha), stage, (double *)nullptr, 0);
    PXSOM_LAUNCH_CHECK("batch_update_kernel");
    return PXSOM_OK;
}

// ------------------------------------------------------------------------------------------------
// labels + per-cluster sums / counts in ONE pass over x (pxsom_assign_sums): for the register-resident shapes the
// accumulating filter (labels, listed rows settled in binary64 inside the launch, per-workgroup binary64 tables) leaves
// [k*c sums | k counts as binary64] in a scratch region behind the assign workspace; a small kernel adds them into the
// caller's tables.  Other shapes: pxsom_assign, then pxsom_cluster_sums.
// ------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void stats_to_tables_kernel(double *__restrict__ stats, int k, int c, double *sums,
                                                              long long *counts)
{
    // (every element is read by exactly one thread, which clears it: the statistics region is left zero, include/pxsom.h)
    for (int e = blockIdx.x * 256 + threadIdx.x; e < k * c + k; e += gridDim.x * 256) {
        if (e < k * c) sums[e] += stats[e];
        else counts[e - k * c] += (long long)stats[e];
        stats[e] = 0.0;
    }
}

// the same, OVERWRITING the caller's tables and forming the means (pxsom_assign_means)
__global__ __launch_bounds__(256) void stats_to_means_kernel(double *__restrict__ stats, int k, int c, double *sums,
                                                             long long *counts, double *means, int *done)
{
    for (int e = blockIdx.x * 256 + threadIdx.x; e < k * c + k; e += gridDim.x * 256) {
        if (e < k * c) {
            const double s = stats[e], cnt = stats[(size_t)k * c + e / c];
            sums[e] = s;
            if (means) means[e] = s / (cnt > 0.0 ? cnt : 1.0);
        } else {
            counts[e - k * c] = (long long)stats[e];
        }
    }
    // the counts are read by the threads of the sums too: the region is cleared by the LAST workgroup to finish (a ticket in the
    // word behind the statistics), so that it is left zero (include/pxsom.h)
    __shared__ int s_last;
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
    __syncthreads();
    if (s_last) {
        for (int e = threadIdx.x; e < k * c + k; e += 256) stats[e] = 0.0;
        if (threadIdx.x == 0) *done = 0;
    }
}
__global__ __launch_bounds__(256) void tables_to_means_kernel(const double *__restrict__ sums, const long long *__restrict__ counts,
                                                              int k, int c, double *means)
{
    for (int e = blockIdx.x * 256 + threadIdx.x; e < k * c; e += gridDim.x * 256) {
        const long long cnt = counts[e / c];
        means[e] = sums[e] / (double)(cnt > 0 ? cnt : 1);
    }
}
}  // namespace

PXSOM_EXPORT size_t pxsom_assign_sums_scratch_bytes(int c, int k)
{
    if (c < 1 || c > PXSOM_MAX_CHANNELS || k < 1 || k > PXSOM_MAX_NODES) return 0;
    return pxsom::align_up((size_t)k * (c + 1) * sizeof(double), 256) + 256;
}

PXSOM_EXPORT size_t pxsom_assign_sums_workspace_bytes(int64_t n, int c, int k)
{
    const size_t a = pxsom_assign_workspace_bytes(n, c, k);
    // [statistics | 256 bytes: the ticket word of the launch that finishes the tables itself] [assign workspace]: the statistics
    // region sits at the START, where it does not move with n (pxsom_assign_sums_scratch_bytes)
    return a ? pxsom_assign_sums_scratch_bytes(c, k) + pxsom::align_up(a, 256) : 0;
}

PXSOM_EXPORT int pxsom_assign_sums(const void *x_dev, int64_t n, int c, int64_t ldx, int dtype, const double *w_dev, int k,
                                   int32_t *labels_dev, double *sums_dev, int64_t *counts_dev, void *workspace_dev,
                                   size_t workspace_bytes, void *stream)
{
    return pxsom_assign_sums_ex(x_dev, n, c, ldx, dtype, w_dev, k, labels_dev, sums_dev, counts_dev, workspace_dev, workspace_bytes, 0, stream);
}

PXSOM_EXPORT int pxsom_assign_sums_ex(const void *x_dev, int64_t n, int c, int64_t ldx, int dtype, const double *w_dev, int k,
                                      int32_t *labels_dev, double *sums_dev, int64_t *counts_dev, void *workspace_dev,
                                      size_t workspace_bytes, int flags, void *stream)
{
    int rc = check_matrix("pxsom_assign_sums", x_dev, n, c, ldx, dtype);
    if (rc) return rc;
    if (k < 1 || k > PXSOM_MAX_NODES)
        return pxsom::fail(PXSOM_ERR_UNSUPPORTED, "pxsom_assign_sums: k=%d outside [1, %d]", k, PXSOM_MAX_NODES);
    if (!w_dev || !sums_dev || !counts_dev || (n > 0 && !labels_dev))
        return pxsom::fail(PXSOM_ERR_INVALID_ARG, "pxsom_assign_sums: null pointer");
    const size_t need = pxsom_assign_sums_workspace_bytes(n, c, k);
    if (!workspace_dev || workspace_bytes < need)
        return pxsom::fail(PXSOM_ERR_WORKSPACE, "pxsom_assign_sums: workspace %zu < %zu bytes", workspace_bytes, need);
    if (n == 0) return PXSOM_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const size_t scratch_bytes = pxsom_assign_sums_scratch_bytes(c, k), stats_bytes = scratch_bytes - 256;
    double *scratch = reinterpret_cast<double *>(workspace_dev);
    void *assign_dev = reinterpret_cast<char *>(workspace_dev) + scratch_bytes;
    const size_t assign_ws = workspace_bytes - scratch_bytes;
    if (!(flags & PXSOM_TABLES_SCRATCH_CLEAN)) PXSOM_HIP_TRY(hipMemsetAsync(scratch, 0, scratch_bytes, st));   // statistics + ticket
    bool fused = false;
    pxsom_bmu::FinishTables fin;
    fin.sums = sums_dev;
    fin.counts = reinterpret_cast<long long *>(counts_dev);
    fin.ticket = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(scratch) + stats_bytes);
    rc = pxsom_bmu::assign_accumulate(x_dev, n, c, ldx, dtype, w_dev, k, labels_dev, scratch, assign_dev, assign_ws, st,
                                      &fused, &fin);
    if (rc) return rc;
    if (fused && fin.done) return PXSOM_OK;   // the last workgroup of the launch added the statistics into the tables (and cleared them)
    if (fused) {
        hipLaunchKernelGGL(stats_to_tables_kernel, dim3((k * (c + 1) + 255) / 256), dim3(256), 0, st, scratch, k, c, sums_dev,
                           reinterpret_cast<long long *>(counts_dev));
        PXSOM_LAUNCH_CHECK("stats_to_tables_kernel");
        return PXSOM_OK;
    }
    rc = pxsom_assign(x_dev, n, c, ldx, dtype, w_dev, k, labels_dev, nullptr, assign_dev, assign_ws, stream);
    if (rc) return rc;
    return pxsom_cluster_sums(x_dev, n, c, ldx, dtype, labels_dev, k, sums_dev, counts_dev, stream);
}

PXSOM_EXPORT int pxsom_assign_means(const void *x_dev, int64_t n, int c, int64_t ldx, int dtype, const double *w_dev, int k,
                                    int32_t *labels_dev, double *sums_dev, int64_t *counts_dev, double *means_dev,
                                    void *workspace_dev, size_t workspace_bytes, void *stream)
{
    return pxsom_assign_means_ex(x_dev, n, c, ldx, dtype, w_dev, k, labels_dev, sums_dev, counts_dev, means_dev, workspace_dev,
                                 workspace_bytes, 0, stream);
}

PXSOM_EXPORT int pxsom_assign_means_ex(const void *x_dev, int64_t n, int c, int64_t ldx, int dtype, const double *w_dev, int k,
                                       int32_t *labels_dev, double *sums_dev, int64_t *counts_dev, double *means_dev,
                                       void *workspace_dev, size_t workspace_bytes, int flags, void *stream)
{
    int rc = check_matrix("pxsom_assign_means", x_dev, n, c, ldx, dtype);
    if (rc) return rc;
    if (k < 1 || k > PXSOM_MAX_NODES)
        return pxsom::fail(PXSOM_ERR_UNSUPPORTED, "pxsom_assign_means: k=%d outside [1, %d]", k, PXSOM_MAX_NODES);
    if (!w_dev || !sums_dev || !counts_dev || (n > 0 && !labels_dev))
        return pxsom::fail(PXSOM_ERR_INVALID_ARG, "pxsom_assign_means: null pointer");
    const size_t need = pxsom_assign_sums_workspace_bytes(n, c, k);
    if (!workspace_dev || workspace_bytes < need)
        return pxsom::fail(PXSOM_ERR_WORKSPACE, "pxsom_assign_means: workspace %zu < %zu bytes", workspace_bytes, need);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const size_t scratch_bytes = pxsom_assign_sums_scratch_bytes(c, k), stats_bytes = scratch_bytes - 256;
    double *scratch = reinterpret_cast<double *>(workspace_dev);
    void *assign_dev = reinterpret_cast<char *>(workspace_dev) + scratch_bytes;
    const size_t assign_ws = workspace_bytes - scratch_bytes;
    const unsigned fgrid = (unsigned)((k * (c + 1) + 255) / 256);
    if (!(flags & PXSOM_TABLES_SCRATCH_CLEAN)) PXSOM_HIP_TRY(hipMemsetAsync(scratch, 0, scratch_bytes, st));   // statistics + ticket
    bool fused = false;
    pxsom_bmu::FinishTables fin;
    fin.sums = sums_dev;
    fin.counts = reinterpret_cast<long long *>(counts_dev);
    fin.means = means_dev;
    fin.overwrite = 1;
    fin.ticket = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(scratch) + stats_bytes);
    if (n > 0) {
        rc = pxsom_bmu::assign_accumulate(x_dev, n, c, ldx, dtype, w_dev, k, labels_dev, scratch, assign_dev, assign_ws, st, &fused,
                                          &fin);
        if (rc) return rc;
    }
    if (fused && fin.done) return PXSOM_OK;   // the last workgroup of the launch wrote the three tables (and cleared the statistics)
    if (fused || n == 0) {   // one launch writes the three tables
        hipLaunchKernelGGL(stats_to_means_kernel, dim3(fgrid), dim3(256), 0, st, scratch, k, c, sums_dev,
                           reinterpret_cast<long long *>(counts_dev), means_dev, reinterpret_cast<int *>(fin.ticket));
        PXSOM_LAUNCH_CHECK("stats_to_means_kernel");
        return PXSOM_OK;
    }
    PXSOM_HIP_TRY(hipMemsetAsync(sums_dev, 0, (size_t)k * c * sizeof(double), st));
    PXSOM_HIP_TRY(hipMemsetAsync(counts_dev, 0, (size_t)k * sizeof(int64_t), st));
    rc = pxsom_assign(x_dev, n, c, ldx, dtype, w_dev, k, labels_dev, nullptr, assign_dev, assign_ws, stream);
    if (rc) return rc;
    rc = pxsom_cluster_sums(x_dev, n, c, ldx, dtype, labels_dev, k, sums_dev, counts_dev, stream);
    if (rc || !means_dev) return rc;
    hipLaunchKernelGGL(tables_to_means_kernel, dim3(fgrid), dim3(256), 0, st, sums_dev, reinterpret_cast<const long long *>(counts_dev), k, c,
                       means_dev);
    PXSOM_LAUNCH_CHECK("tables_to_means_kernel");
    return PXSOM_OK;
}

// codebooks the accumulating filter prepares for itself inside its own launch (register-resident shapes)
static bool self_preparing_shape(int c, int k)
{
    const pxsom_bmu::Layout L = pxsom_bmu::make_layout(0, c, k);
    return c % 2 == 0 && L.nch == 1 && L.nb == 7 && (k - 16 * (L.nb - 1) + 3) / 4 == 1;
}

// One mini-batch step's accumulation half: zero the statistics, BMU of every row, per-BMU sums.
// stats_dev = [k*c sums | k counts], all binary64 (counts are exact integers below 2^53), so the
// multi-GPU all-reduce is a single sum over one buffer.
// flags & PXSOM_ACC_PREPARED: pxsom_batch_update_prepare already cleared stats_dev (and prepared the workspace
// for w_dev where the shape needs one).
static int batch_accumulate_impl(const void *x_dev, int64_t n, int c, int64_t ldx, int dtype, const double *w_dev, int k,
                                 int32_t *labels_dev, double *stats_dev, void *workspace_dev, size_t workspace_bytes, int flags,
                                 void *stream, double qmagic);

PXSOM_EXPORT int pxsom_batch_accumulate(const void *x_dev, int64_t n, int c, int64_t ldx, int dtype,
                                        const double *w_dev, int k, int32_t *labels_dev, double *stats_dev,
                                        void *workspace_dev, size_t workspace_bytes, int flags, void *stream)
{
    return batch_accumulate_impl(x_dev, n, c, ldx, dtype, w_dev, k, labels_dev, stats_dev, workspace_dev, workspace_bytes, flags,
                                 stream, 0.0);
}

// qmagic != 0: binary64 rows rounded to the run's quantum as they join the statistics (the one-launch accumulating filter
// does not know the rounding: search and sums run as two kernels then)
static int batch_accumulate_impl(const void *x_dev, int64_t n, int c, int64_t ldx, int dtype, const double *w_dev, int k,
                                 int32_t *labels_dev, double *stats_dev, void *workspace_dev, size_t workspace_bytes, int flags,
                                 void *stream, double qmagic)
{
    if (dtype != PXSOM_F64) qmagic = 0.0;
    if (!stats_dev || k < 1 || k > PXSOM_MAX_NODES || c < 1 || c > PXSOM_MAX_CHANNELS)
        return pxsom::fail(PXSOM_ERR_INVALID_ARG, "pxsom_batch_accumulate: bad statistics buffer / shape");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const bool cleared = (flags & PXSOM_ACC_PREPARED) != 0;
    if (!cleared) PXSOM_HIP_TRY(hipMemsetAsync(stats_dev, 0, (size_t)k * (c + 1) * sizeof(double), st));
    // fused route (register-resident filter shapes): ONE launch prepares the codebook, labels every row,
    // settles the listed rows and accumulates -- one pass over x
    bool fused = false;
    int rc = PXSOM_OK;
    if (qmagic == 0.0)
        rc = pxsom_bmu::assign_accumulate(x_dev, n, c, ldx, dtype, w_dev, k, labels_dev, stats_dev, workspace_dev,
                                          workspace_bytes, st, &fused);
    if (fused) return rc;
    if (n == 0) return PXSOM_OK;
    rc = check_matrix("pxsom_batch_accumulate", x_dev, n, c, ldx, dtype);
    if (rc) return rc;
    // a workspace prepared by update_prepare exists only for shapes that are not self-preparing
    if (cleared && !self_preparing_shape(c, k)) {
        if (!w_dev || !labels_dev) return pxsom::fail(PXSOM_ERR_INVALID_ARG, "pxsom_batch_accumulate: null pointer");
        rc = pxsom_bmu::assign_prepared(x_dev, n, c, ldx, dtype, w_dev, k, labels_dev, workspace_dev, workspace_bytes, st);
    } else {
        rc = pxsom_assign(x_dev, n, c, ldx, dtype, w_dev, k, labels_dev, nullptr, workspace_dev, workspace_bytes, stream);
    }
    if (rc) return rc;
    double *sums = stats_dev;
    int64_t *counts = reinterpret_cast<int64_t *>(stats_dev + (size_t)k * c);
    PXSOM_DISPATCH_DTYPE(dtype, x_dev, xp, (cluster_sums_typed<T, true>(xp, n, c, ldx, labels_dev, k, sums, counts, st, qmagic)));
}

// The update half of a mini-batch step plus what the NEXT pxsom_batch_accumulate(PXSOM_ACC_PREPARED) relies on:
// codebook update from the (all-reduced) statistics in stats_dev; stats_next_dev -- the buffer the next
// accumulate will fill -- cleared by the same launch (pass the other one of two alternating buffers; with
// stats_next_dev == stats_dev or NULL the buffer is cleared by a separate fill); and, for codebook shapes the
// accumulating filter does not prepare itself, the workspace prepared for the new codebook.
// (A single-workgroup fusion of update and prep was measured slower: its window sums are LDS-bandwidth bound
// on one CU, 9-14 us at radius 6; so was a last-workgroup-runs-prep variant.)
PXSOM_EXPORT int pxsom_batch_update_prepare(double *w_dev, int xdim, int ydim, int c, double *stats_dev,
                                            double *stats_next_dev, double thr, double alpha,
                                            void *workspace_dev, size_t workspace_bytes, void *stream)
{
    if (xdim < 1 || ydim < 1 || (int64_t)xdim * ydim > PXSOM_MAX_NODES || c < 1 || c > PXSOM_MAX_CHANNELS)
        return pxsom::fail(PXSOM_ERR_UNSUPPORTED, "pxsom_batch_update_prepare: shape %dx%d x %d", xdim, ydim, c);
    if (!w_dev || !stats_dev) return pxsom::fail(PXSOM_ERR_INVALID_ARG, "pxsom_batch_update_prepare: null pointer");
    const int k = xdim * ydim;
    const bool needs_ws = !self_preparing_shape(c, k);
    if (needs_ws && (!workspace_dev || workspace_bytes < pxsom_assign_workspace_bytes(0, c, k)))
        return pxsom::fail(PXSOM_ERR_WORKSPACE, "pxsom_batch_update_prepare: workspace %zu bytes too small",
                           workspace_bytes);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int nstats = k * (c + 1);
    const bool other = stats_next_dev && stats_next_dev != stats_dev;
    const size_t stage_bytes = (size_t)nstats * sizeof(double);
    const int stage = stage_bytes <= 60 * 1024;
    hipLaunchKernelGGL(batch_update_kernel, dim3(k), dim3(256), stage ? stage_bytes : 0, st, w_dev, xdim, ydim, c,
                       stats_dev, stats_dev + (size_t)k * c, thr, 1.0 - alpha, pxsom_bmu::batch_gain_saturation(1.0 - alpha), stage, other ? stats_next_dev : nullptr,
                       other ? nstats : 0);
    PXSOM_LAUNCH_CHECK("batch_update_kernel");
    if (!other) PXSOM_HIP_TRY(hipMemsetAsync(stats_dev, 0, stage_bytes, st));
    if (!needs_ws) return PXSOM_OK;
    return pxsom_bmu::prepare_only(w_dev, c, k, workspace_dev, workspace_bytes, nullptr, st);
}

// ------------------------------------------------------------------------------------------------
// Batch training pass driven from ONE call (pxsom_batch_train_steps): the host loop over the mini-batch steps
// lives here, not in Python.  Register-resident shapes on a 10 x 10 grid take one launch per step (batch_step_kernel,
// pxsom_batch_step.hip: pending update + prep + filter + table + flush); other shapes run update-and-prepare /
// filter / exact / cluster sums per step.  Both keep the same state:
//   wbuf[g % 2]        W_g, the codebook step g searches with          (two buffers alternate)
//   ring[g % 3]        statistics of step g; ring[(g+1) % 3] is cleared by step g
// so a multi-rank run all-reduces ring[g % 3] right behind step g (comm != NULL: enqueued here, pxsom_comm.hip; or the
// caller runs one step per call and all-reduces in between).
// ------------------------------------------------------------------------------------------------
namespace {

// (thr, alpha) of the online schedule at position pos / span of the run (pos = rows presented before the step, in
// phases: orc_som_batch_sched)
inline void batch_schedule(int64_t pos, int64_t span, double a0, double a1, double r0, double r1, double *thr, double *alpha)
{
    double t = r0 - (r0 - r1) * (double)pos / (double)span;
    if (t < 1.0) t = 0.5;
    *thr = t;
    *alpha = a0 - (a0 - a1) * (double)pos / (double)span;
}

// A pass's schedule: row i belongs to phase i % phases, step g takes the phases [edges[g], edges[g+1]).
struct Sched {
    int phases, steps;
    const int32_t *edges;   // host, [steps + 1]
    int e0(int g) const { return edges[g]; }
    int width(int g) const { return edges[g + 1] - edges[g]; }
    int64_t rows(int64_t n, int g) const
    {
        const int64_t full = n / phases, rem = n % phases;
        const int64_t part = std::min<int64_t>(std::max<int64_t>(rem - e0(g), 0), width(g));
        return full * width(g) + part;
    }
    int64_t offset(int64_t n, int g) const   // rows of the steps before g
    {
        const int64_t full = n / phases, rem = n % phases;
        return full * e0(g) + std::min<int64_t>(e0(g), rem);
    }
    int64_t rows_max(int64_t n) const
    {
        int64_t m = 0;
        for (int g = 0; g < steps; g++) m = std::max(m, rows(n, g));
        return m;
    }
    bool any_wide() const
    {
        for (int g = 0; g < steps; g++)
            if (width(g) > 1) return true;
        return false;
    }
    int64_t pos(int gg) const { return (int64_t)(gg / steps) * phases + e0(gg % steps); }   // of global step gg
};

int check_sched(const char *fn, int phases, const int32_t *edges, int steps)
{
    if (phases < 1 || steps < 1 || steps > PXSOM_MAX_SCHED_STEPS || !edges || edges[0] != 0 || edges[steps] != phases)
        return pxsom::fail(PXSOM_ERR_INVALID_ARG, "%s: schedule needs 1 <= steps <= %d, edges[0] == 0, edges[steps] == phases", fn,
                           PXSOM_MAX_SCHED_STEPS);
    for (int g = 0; g < steps; g++)
        if (edges[g + 1] < edges[g]) return pxsom::fail(PXSOM_ERR_INVALID_ARG, "%s: schedule edges must not decrease", fn);
    return PXSOM_OK;
}

// Steps that take several phases (width > 1) on shapes outside the fused kernel: their rows are gathered ONCE per run
// into step-contiguous order (the generic search / exact / sums kernels take plain strided matrices).  One work item
// per (destination row, 16 / 8 / 4 / 2-byte chunk); the step of a destination row by binary search over the
// closed-form offsets.
struct SchedArg {
    int phases, steps;
    int edges[PXSOM_MAX_SCHED_STEPS + 1];
};

template <typename V>
__global__ __launch_bounds__(256) void gather_steps_kernel(const V *__restrict__ x, int64_t n, int cpr, int64_t ldx_v,
                                                           V *__restrict__ out, SchedArg s)
{
    const int64_t full = n / s.phases, rem = n % s.phases;
    auto off = [&](int g) { return full * s.edges[g] + min((int64_t)s.edges[g], rem); };
    const int64_t items = n * cpr;
    for (int64_t it = (int64_t)blockIdx.x * 256 + threadIdx.x; it < items; it += (int64_t)gridDim.x * 256) {
        const int64_t d = it / cpr;
        const int ch = (int)(it - d * cpr);
        int lo = 0, hi = s.steps;   // the step with off(lo) <= d < off(lo + 1)
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (off(mid) <= d) lo = mid;
            else hi = mid;
        }
        const int64_t r = d - off(lo);
        const int w = s.edges[lo + 1] - s.edges[lo];
        const int64_t src = (r / w) * s.phases + s.edges[lo] + (r % w);
        out[d * cpr + ch] = x[src * ldx_v + ch];
    }
}

template <typename T>
int launch_gather(const T *x, int64_t n, int c, int64_t ldx, T *out, const Sched &sc, hipStream_t st)
{
    SchedArg a;
    a.phases = sc.phases;
    a.steps = sc.steps;
    for (int g = 0; g <= sc.steps; g++) a.edges[g] = sc.edges[g];
    const size_t rb = (size_t)c * sizeof(T), lb = (size_t)ldx * sizeof(T);
    const uintptr_t ax = reinterpret_cast<uintptr_t>(x), ao = reinterpret_cast<uintptr_t>(out);
    const int64_t grid_max = (int64_t)pxsom::device_cu_count() * 16;
    auto go = [&](auto tag) {
        typedef decltype(tag) V;
        const int cpr = (int)(rb / sizeof(V));
        const int64_t grid = std::min<int64_t>((n * cpr + 255) / 256, grid_max);
        hipLaunchKernelGGL(gather_steps_kernel<V>, dim3((unsigned)std::max<int64_t>(grid, 1)), dim3(256), 0, st,
                           reinterpret_cast<const V *>(x), n, cpr, (int64_t)(lb / sizeof(V)), reinterpret_cast<V *>(out), a);
    };
    if (rb % 16 == 0 && lb % 16 == 0 && ax % 16 == 0 && ao % 16 == 0) go(uint4{});
    else if (rb % 8 == 0 && lb % 8 == 0 && ax % 8 == 0 && ao % 8 == 0) go(uint2{});
    else if (rb % 4 == 0 && lb % 4 == 0 && ax % 4 == 0 && ao % 4 == 0) go((unsigned)0);
    else go((unsigned short)0);
    PXSOM_LAUNCH_CHECK("gather_steps_kernel");
    return PXSOM_OK;
}

// The run's centring vector for the one-launch step's filter (AssignHdr::mu_s, DESIGN.md "K7 centring"): the mean of the
// codebook the run starts from, per channel, in binary32.  Any vector keeps the search exact; this one stays close to the
// nodes' mean for the whole run (they follow the data), so the steps need no reduction of their own for it.
__global__ __launch_bounds__(1024) void centring_vector_kernel(const double *__restrict__ w, int k, int c, float *__restrict__ mu32,
                                                               double *__restrict__ zero_out, int zero_count, double *__restrict__ copy_out)
{
    // 1024 threads clear and copy (a 100 x 100 codebook on 256 threads was 40 dependent load -> store trips: 22 us of config 4's
    // pass), the first 256 form the means
    for (int e = threadIdx.x; e < zero_count; e += 1024) zero_out[e] = 0.0;   // the first step's statistics buffer (no memset launch)
    if (copy_out) {   // W_0 handed over by the caller: into the run's codebook buffer (no copy launch in front of the pass)
        for (int e0 = threadIdx.x; e0 < k * c; e0 += 4 * 1024) {
            double v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = w[e0 + u * 1024 < k * c ? e0 + u * 1024 : 0];
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (e0 + u * 1024 < k * c) copy_out[e0 + u * 1024] = v[u];
        }
    }
    __shared__ float s_m[pxsom_bmu::kFilterMaxChannels];
    if (threadIdx.x < pxsom_bmu::kFilterMaxChannels) s_m[threadIdx.x] = 0.f;
    __syncthreads();
    // `parts` adjacent lanes share a channel (8 for c <= 32, 2 for c <= 128): each sums every parts-th node with its loads in
    // flight eight at a time -- a lane walking 50 nodes one L2 round trip after the other made this launch 16 us
    if (threadIdx.x < 256) {
        const int cp = c <= 32 ? 32 : (c <= 64 ? 64 : 128), parts = 256 / cp;
        const int j = threadIdx.x / parts, part = threadIdx.x % parts;
        double sum = 0.0;
        if (j < c) {
            for (int n0 = part; n0 < k; n0 += 8 * parts) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = n0 + u * parts < k ? w[(size_t)(n0 + u * parts) * c + j] : 0.0;
#pragma unroll
                for (int u = 0; u < 8; u++) sum += v[u];
            }
        }
        for (int d = 1; d < parts; d *= 2) sum += __shfl_xor(sum, d);
        float m = (float)(sum / (double)k);
        if (!(j < c && fabsf(m) <= 3.0e38f)) m = 0.f;   // a non-finite codebook: not centred (every row is listed anyway)
        if (part == 0 && j < pxsom_bmu::kFilterMaxChannels) {
            mu32[j] = m;
            s_m[j] = m;
        }
    }
    __syncthreads();
    // word 128: the vector's norm (the steps cap their power-of-two scale with it: pxsom_batch_step.hip)
    if (threadIdx.x < 64) {
        const double a = (double)s_m[threadIdx.x], b = (double)s_m[threadIdx.x + 64];
        double n2 = a * a + b * b;
        for (int d = 1; d < 64; d *= 2) n2 += __shfl_xor(n2, d);
        if (threadIdx.x == 0) mu32[pxsom_bmu::kFilterMaxChannels] = (float)sqrt(n2);
    }
}

struct TrainWs {
    size_t assign_ws, off_labels, off_mu, off_gather, total;
};
inline TrainWs train_ws(int64_t n, int c, int k, size_t esize, const Sched &sc)
{
    TrainWs w;
    const int64_t rmax = sc.rows_max(n);
    w.assign_ws = pxsom_assign_workspace_bytes(rmax, c, k);
    w.off_labels = pxsom::align_up(w.assign_ws, 256);
    w.off_mu = w.off_labels + pxsom::align_up((size_t)(rmax > 0 ? rmax : 1) * sizeof(int32_t), 256);
    w.off_gather = w.off_mu + 1024;   // 129 floats: the run's centring vector (c <= 128) and its norm
    w.total = w.off_gather + (sc.any_wide() ? pxsom::align_up((size_t)(n > 0 ? n : 1) * c * esize, 256) : 0);
    return w;
}
}  // namespace
namespace pxsom {
int comm_allreduce_sum_f64(pxsom_comm *c, double *buf, size_t count, hipStream_t st);   // pxsom_comm.hip
}
namespace {

template <typename T>
int train_steps_typed(const T *x, int64_t n, int c, int64_t ldx, int dtype, double *wbuf, double *ring, int xdim,
                      int ydim, const Sched &sc, int g_begin, int g_end, int num_passes, double a0, double a1, double r0,
                      double r1, double sum_quantum, char *ws, size_t ws_bytes, int flags, pxsom_comm *comm, hipStream_t st,
                      const double *w0 = nullptr)
{
    // binary64 rows of a reproducible run: (v + qmagic) - qmagic rounds v to a multiple of the quantum
    const double qmagic = (sizeof(T) == 8 && sum_quantum > 0.0) ? 6755399441055744.0 /* 1.5 * 2^52 */ * sum_quantum : 0.0;
    const int k = xdim * ydim;
    const size_t nstats = (size_t)k * (c + 1), nw = (size_t)k * c;
    const TrainWs tw = train_ws(n, c, k, sizeof(T), sc);
    const size_t assign_ws = tw.assign_ws;
    int32_t *labels = reinterpret_cast<int32_t *>(ws + tw.off_labels);
    T *xg = reinterpret_cast<T *>(ws + tw.off_gather);
    const int64_t span = (int64_t)num_passes * sc.phases;
    constexpr int tpw = 0;   // 16-row tiles per wave of the fused step: by step size (launch_step)
    // which route a step takes (one decision per run: every step of a shape shares it, and so do all ranks -- the fused
    // kernel needs rows >= 1, which a rank with a short shard may not have, so empty steps are allowed there)
    const bool fused_shape = !(flags & PXSOM_TRAIN_UNFUSED) &&
                             pxsom_bmu::step_fused_shape<T>(x, 1, c, ldx, xdim, ydim, (int64_t)sc.phases * ldx);
    float *mu32 = reinterpret_cast<float *>(ws + tw.off_mu);
    // rows of 2-byte floats on the generic route keep the uncentred two-term split (pxsom_assign_filter.hip)
    const bool centred_run = fused_shape || (sizeof(T) != 2 && c <= pxsom_bmu::kFilterMaxChannels && !(flags & PXSOM_TRAIN_UNFUSED));
    if (g_begin == 0) {   // the first step's statistics buffer; every later one is cleared by the step before it
        // (w0: the codebook the run starts from, where the caller holds it -- copied into wbuf[0] by the launch that is there anyway)
        if (centred_run) {
            hipLaunchKernelGGL(centring_vector_kernel, dim3(1), dim3(1024), 0, st, w0 ? w0 : wbuf, k, c, mu32, ring, (int)nstats,
                               w0 ? wbuf : (double *)nullptr);
            PXSOM_LAUNCH_CHECK("centring_vector_kernel");
        } else {
            if (w0) PXSOM_HIP_TRY(hipMemcpyAsync(wbuf, w0, nw * sizeof(double), hipMemcpyDeviceToDevice, st));
            PXSOM_HIP_TRY(hipMemsetAsync(ring, 0, nstats * sizeof(double), st));
        }
    }
    // Round 6: where every kernel of the generic route takes row views (pxsom_common.h RowView: more than 64 channels of binary32 /
    // binary16 rows, contiguous in the caller's matrix) the steps read their rows where they lie -- no gathered copy of the matrix
    // at the head of every pass (config 4: 217 us of 2.0 ms, 800 MB of traffic)
    const bool viewed = !fused_shape && sc.any_wide() && c > 32 && c <= pxsom_bmu::kFilterMaxChannels && sums_take_views<T>(x, c, ldx, k);
    const bool gathered = !fused_shape && sc.any_wide() && !viewed;
    if (gathered && g_begin == 0 && n > 0) {
        int rc = launch_gather<T>(x, n, c, ldx, xg, sc, st);
        if (rc) return rc;
    }
    // coefficients of the fused kernels' rigorous |score - exact| bound (DESIGN.md "K7 error bound"): tol = 2 * 1.25 * E (+ 2^-24:
    // the rounding of the centred row, x' = fl(x * scale - mu_s)).  Index bits packed into the scores: 5 in the one-launch step
    // (round 6: the lane group travels beside the scores).
    const float fused_tol_rel = (float)(2.5 * (ldexp(1.0, -(23 - 5)) + pxsom_bmu::filter_accum_units(c, 3) * ldexp(1.0, -24) + ldexp(1.0, -19) +
                                               ldexp(1.0, -23) + ldexp(1.0, -24)));
    const float fused_tol_abs = (float)pxsom_bmu::filter_tol_abs(c);
    constexpr bool no_centre = false;
    // (Rounds 4 - 5 carried an opt-in persistent launch for the BMU-only tail on one XCD, pxsom_batch_tail.hip: 12.2 us per step
    // against 9.8 for the launches, profiles/r04/tail_phase_timing.txt; removed in round 6 with its flag.)
    const int g_tail = g_end;
    // The exchange inside the step launches (round 5; a peer-to-peer communicator with pxsom_comm_p2p_set_fused, the fused 10 x 10 step):
    // step gg's last workgroup hands this rank's statistics to every rank, step gg + 1 adds the ranks' slots in rank order while
    // it applies the pending update -- no all-reduce launch between two steps.  The last step of the call keeps the separate
    // all-reduce: the next call (or the final update) reads the ring.  Every rank takes the same decision (same environment,
    // same communicator kind, agreed route).
    pxsom::FusedXch fxch;
    bool fused_xch = false;
    {
        if (comm && fused_shape && g_tail - g_begin >= 2)   // (false unless the communicator is peer-to-peer with its fused switch on)
            fused_xch = pxsom::comm_fused_begin(comm, g_tail - g_begin - 1, nstats, &fxch);
    }
    for (int gg = g_begin; gg < g_tail; gg++) {
        const int g = gg % sc.steps;
        const int64_t rows = sc.rows(n, g);
        const int wd = sc.width(g);
        // the step's rows as a strided matrix: phase view (one phase), or its slice of the gathered copy
        const T *xv = gathered ? xg + (size_t)sc.offset(n, g) * c : x + (size_t)sc.e0(g) * ldx;
        const int64_t ldv = gathered ? c : ((viewed && wd > 1) ? ldx : ldx * sc.phases);
        const pxsom::RowViewScope view_scope((viewed && wd > 1) ? pxsom::make_row_view(wd, (int64_t)sc.phases * ldx) : pxsom::RowView{});
        if (rows >= (int64_t)1 << 31) return pxsom::fail(PXSOM_ERR_UNSUPPORTED, "pxsom_batch_train: a step of %lld rows", (long long)rows);
        double *w_prev = wbuf + (size_t)((gg + 1) % 2) * nw, *w_cur = wbuf + (size_t)(gg % 2) * nw;
        double *s_prev = ring + (size_t)((gg + 2) % 3) * nstats, *s_cur = ring + (size_t)(gg % 3) * nstats,
               *s_next = ring + (size_t)((gg + 1) % 3) * nstats;
        double thr = 0.0, alpha = 0.0;
        if (gg > 0) batch_schedule(sc.pos(gg - 1), span, a0, a1, r0, r1, &thr, &alpha);
        if (fused_shape) {
            pxsom_bmu::StepArgs sa;
            sa.w_in = gg > 0 ? w_prev : w_cur;
            sa.w_out = w_cur;
            sa.stats_prev = s_prev;
            sa.stats_zero = s_next;
            sa.zero_count = (int)nstats;
            sa.has_update = gg > 0 ? 1 : 0;
            sa.thr = thr;
            sa.q = 1.0 - alpha;
            sa.sat = pxsom_bmu::batch_gain_saturation(sa.q);
            sa.tol_rel = fused_tol_rel;
            sa.tol_abs = fused_tol_abs;
            sa.mu32 = no_centre ? nullptr : mu32;
            sa.group_w = wd > 1 ? wd : 1;
            sa.group_stride = (int64_t)sc.phases * ldx;
            sa.qmagic = qmagic;
            if (fused_xch) {
                const int i = gg - g_begin;                       // the exchange behind step gg has epoch base + i + 1
                sa.xch_peers = fxch.peers;
                sa.xch_ticket = fxch.ticket;
                sa.xch_nranks = fxch.nranks;
                sa.xch_rank = fxch.rank;
                sa.xch_max_count = fxch.max_count;
                sa.xch_wait = i > 0 ? fxch.epoch_base + (unsigned long long)i : 0ull;
                sa.xch_signal = gg + 1 < g_tail ? fxch.epoch_base + (unsigned long long)i + 1ull : 0ull;
            }
            // (an empty step -- a rank whose shard is shorter than the schedule -- still launches: the update, the
            // clearing of the next buffer and W_g are the kernel's, and every rank must take the same route)
            int rc = pxsom_bmu::launch_batch_step<T>(x + (size_t)sc.e0(g) * ldx, rows, c, wd > 1 ? ldx : ldx * sc.phases,
                                                     s_cur, sa, tpw, st);
            if (rc) return rc;
            if (comm && (!fused_xch || gg + 1 == g_tail) && (rc = pxsom::comm_allreduce_sum_f64(comm, s_cur, nstats, st))) return rc;
            continue;
        }
        // small steps (<= 16 K rows) of codebooks up to 256 nodes x 128 channels: ONE launch (update, fragments, search, exact
        // settle, statistics: pxsom_batch_step_wide.hip) instead of the four or five below -- the BMU-only steps (threshold
        // pinned at 0.5) on any grid, the windowed ones on grids up to 16 x 16
        if constexpr (sizeof(T) >= 4) {
            const bool bmu_only = gg > 0 && thr == 0.5;
            constexpr int64_t win_cap = PXSOM_WIDE_WIN_CAP;   // (rows of a windowed step the wide one-launch kernel takes)
            if (!(flags & PXSOM_TRAIN_UNFUSED) && rows <= pxsom_bmu::step_wide_max_rows() && pxsom_bmu::step_wide_shape<T>(c, k) &&
                (bmu_only || (rows <= win_cap && pxsom_bmu::step_wide_windowed(xdim, ydim, c)))) {
                pxsom_bmu::StepArgs sa;
                sa.w_in = gg > 0 ? w_prev : w_cur;
                sa.w_out = gg > 0 ? w_cur : nullptr;
                sa.stats_prev = s_prev;
                sa.stats_zero = s_next;
                sa.zero_count = (int)nstats;
                sa.has_update = gg > 0 ? 1 : 0;
                sa.thr = thr;
                sa.q = 1.0 - alpha;
            sa.sat = pxsom_bmu::batch_gain_saturation(sa.q);
                sa.mu32 = (centred_run && !no_centre) ? mu32 : nullptr;
                // (5 index bits in the scores -- 6 from 129 nodes on --, three-term split, centred rows)
                sa.tol_rel = (float)(2.5 * (ldexp(1.0, -(23 - (k > 128 ? 6 : 5))) + pxsom_bmu::filter_accum_units_split(c, 3) * ldexp(1.0, -24) + ldexp(1.0, -19) + ldexp(1.0, -23) +
                                            ldexp(1.0, -24)));
                sa.tol_abs = fused_tol_abs;
                sa.qmagic = qmagic;
                int rc = pxsom_bmu::launch_batch_step_wide<T>(xv, rows, c, ldv, xdim, ydim, s_cur, sa, st);
                if (rc) return rc;
                if (comm && (rc = pxsom::comm_allreduce_sum_f64(comm, s_cur, nstats, st))) return rc;
                continue;
            }
        }
        // codebooks the all-in-one kernel cannot hold (K = 400, or C > 32): ONE launch applies the pending update and
        // prepares the assign workspace for W_g (copy + update + clears + prep before), then search / exact / sums
        if (!(flags & PXSOM_TRAIN_UNFUSED)) {
            const int npk = pxsom_bmu::packed_rows_ok<T>(xv, ldv) ? pxsom_bmu::packed_k(c, k, sizeof(T) == 2) : 0;
            const pxsom_bmu::Layout L = pxsom_bmu::make_layout(rows, c, k, npk);
            pxsom_bmu::StepArgs sa;
            sa.w_in = gg > 0 ? w_prev : w_cur;
            sa.w_out = gg > 0 ? w_cur : nullptr;
            sa.stats_prev = s_prev;
            sa.stats_zero = s_next;
            sa.zero_count = (int)nstats;
            sa.has_update = gg > 0 ? 1 : 0;
            sa.thr = thr;
            sa.q = 1.0 - alpha;
            sa.sat = pxsom_bmu::batch_gain_saturation(sa.q);
            // the generic filter is centred on the run's vector too (binary32 / binary64 rows): + 2^-24, the rounding of
            // x' = fl(x * scale - mu_s)
            sa.mu32 = (centred_run && !no_centre && npk == 0) ? mu32 : nullptr;
            sa.tol_rel = (float)(2.5 * (ldexp(1.0, -(23 - L.idx_bits)) + pxsom_bmu::filter_accum_units_for(c, 3, npk) * ldexp(1.0, -24) + ldexp(1.0, -19) +
                                        ldexp(1.0, -23) + (sa.mu32 ? ldexp(1.0, -24) : 0.0)));
            sa.tol_abs = (float)pxsom_bmu::filter_tol_abs(c);
            int rc = PXSOM_OK;
            if (pxsom_bmu::launch_update_prepare(sa, xdim, ydim, c, ws, L, st, &rc)) {
                if (rc) return rc;
                if (rows > 0) {
                    rc = pxsom_bmu::assign_prepared(xv, rows, c, ldv, dtype, w_cur, k, labels, ws, assign_ws, st, npk);
                    if (rc) return rc;
                    rc = cluster_sums_typed<T, true>(xv, rows, c, ldv, labels, k, s_cur, reinterpret_cast<int64_t *>(s_cur + nw), st, qmagic);
                    if (rc) return rc;
                }
                if (comm && (rc = pxsom::comm_allreduce_sum_f64(comm, s_cur, nstats, st))) return rc;
                continue;
            }
        }
        if (gg > 0) {
            PXSOM_HIP_TRY(hipMemcpyAsync(w_cur, w_prev, nw * sizeof(double), hipMemcpyDeviceToDevice, st));
            int rc = pxsom_batch_update(w_cur, xdim, ydim, c, s_prev, s_prev + nw, thr, alpha, st);
            if (rc) return rc;
        }
        PXSOM_HIP_TRY(hipMemsetAsync(s_next, 0, nstats * sizeof(double), st));
        int rc = batch_accumulate_impl(xv, rows, c, ldv, dtype, w_cur, k, labels, s_cur, ws, assign_ws, 0, st, qmagic);
        if (rc) return rc;
        if (comm && (rc = pxsom::comm_allreduce_sum_f64(comm, s_cur, nstats, st))) return rc;
    }
    (void)ws_bytes;
    return PXSOM_OK;
}

inline std::vector<int32_t> equal_edges(int m)
{
    std::vector<int32_t> e((size_t)m + 1);
    for (int t = 0; t <= m; t++) e[(size_t)t] = t;
    return e;
}

}  // namespace

PXSOM_EXPORT size_t pxsom_batch_train_sched_workspace_bytes(int64_t n, int c, int k, int dtype, int phases,
                                                            const int32_t *edges, int steps_per_pass)
{
    if (n < 0 || c < 1 || c > PXSOM_MAX_CHANNELS || k < 1 || k > PXSOM_MAX_NODES || !pxsom::dtype_ok(dtype)) return 0;
    if (check_sched("pxsom_batch_train_sched_workspace_bytes", phases, edges, steps_per_pass)) return 0;
    const Sched sc{phases, steps_per_pass, edges};
    return train_ws(n, c, k, dtype == PXSOM_F64 ? 8 : (dtype == PXSOM_F32 ? 4 : 2), sc).total;
}

PXSOM_EXPORT size_t pxsom_batch_train_workspace_bytes(int64_t n, int batch_steps, int c, int k)
{
    if (batch_steps < 1 || batch_steps > PXSOM_MAX_SCHED_STEPS) return 0;
    const std::vector<int32_t> e = equal_edges(batch_steps);
    return pxsom_batch_train_sched_workspace_bytes(n, c, k, PXSOM_F64, batch_steps, e.data(), batch_steps);
}

PXSOM_EXPORT int pxsom_batch_train_sched(const void *x_dev, int64_t n, int c, int64_t ldx, int dtype, double *wbuf_dev,
                                         double *stats_ring_dev, int xdim, int ydim, int phases, const int32_t *edges,
                                         int steps_per_pass, int g_begin, int g_end, int num_passes, double a0, double a1,
                                         double r0, double r1, double sum_quantum, void *workspace_dev, size_t workspace_bytes,
                                         int flags, pxsom_comm *comm, void *stream)
{
    return pxsom_batch_train_sched_from(x_dev, n, c, ldx, dtype, nullptr, wbuf_dev, stats_ring_dev, xdim, ydim, phases, edges, steps_per_pass,
                                        g_begin, g_end, num_passes, a0, a1, r0, r1, sum_quantum, workspace_dev, workspace_bytes, flags, comm,
                                        stream);
}

PXSOM_EXPORT int pxsom_batch_train_sched_from(const void *x_dev, int64_t n, int c, int64_t ldx, int dtype, const double *w0_dev,
                                              double *wbuf_dev, double *stats_ring_dev, int xdim, int ydim, int phases,
                                              const int32_t *edges, int steps_per_pass, int g_begin, int g_end, int num_passes,
                                              double a0, double a1, double r0, double r1, double sum_quantum, void *workspace_dev,
                                              size_t workspace_bytes, int flags, pxsom_comm *comm, void *stream)
{
    int rc = check_matrix("pxsom_batch_train_sched", x_dev, n, c, ldx, dtype);
    if (rc) return rc;
    {
        int qe = 0;
        if (!(sum_quantum >= 0.0) || (sum_quantum > 0.0 && frexp(sum_quantum, &qe) != 0.5))
            return pxsom::fail(PXSOM_ERR_INVALID_ARG, "pxsom_batch_train_sched: sum_quantum must be 0 or a power of two");
    }
    if (rc) return rc;
    if (xdim < 1 || ydim < 1 || (int64_t)xdim * ydim > PXSOM_MAX_NODES)
        return pxsom::fail(PXSOM_ERR_UNSUPPORTED, "pxsom_batch_train_sched: grid %dx%d outside [1, %d] nodes", xdim, ydim,
                           PXSOM_MAX_NODES);
    if ((rc = check_sched("pxsom_batch_train_sched", phases, edges, steps_per_pass))) return rc;
    if (num_passes < 1 || g_begin < 0 || g_end < g_begin || (int64_t)g_end > (int64_t)num_passes * steps_per_pass)
        return pxsom::fail(PXSOM_ERR_INVALID_ARG, "pxsom_batch_train_sched: steps [%d, %d) of %d passes x %d", g_begin, g_end,
                           num_passes, steps_per_pass);
    if (!wbuf_dev || !stats_ring_dev) return pxsom::fail(PXSOM_ERR_INVALID_ARG, "pxsom_batch_train_sched: null pointer");
    const size_t need = pxsom_batch_train_sched_workspace_bytes(n, c, xdim * ydim, dtype, phases, edges, steps_per_pass);
    if (!workspace_dev || workspace_bytes < need)
        return pxsom::fail(PXSOM_ERR_WORKSPACE, "pxsom_batch_train_sched: workspace %zu < %zu bytes", workspace_bytes, need);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const Sched sc{phases, steps_per_pass, edges};
    PXSOM_DISPATCH_DTYPE(dtype, x_dev, xp,
                         train_steps_typed<T>(xp, n, c, ldx, dtype, wbuf_dev, stats_ring_dev, xdim, ydim, sc, g_begin, g_end,
                                              num_passes, a0, a1, r0, r1, sum_quantum, reinterpret_cast<char *>(workspace_dev),
                                              workspace_bytes, flags, comm, st, g_begin == 0 ? w0_dev : nullptr));
}

PXSOM_EXPORT int pxsom_batch_train_steps(const void *x_dev, int64_t n, int c, int64_t ldx, int dtype, double *wbuf_dev,
                                         double *stats_ring_dev, int xdim, int ydim, int batch_steps, int g_begin,
                                         int g_end, int total_steps, double a0, double a1, double r0, double r1,
                                         void *workspace_dev, size_t workspace_bytes, int flags, void *stream)
{
    return pxsom_batch_train_steps_sharded(x_dev, n, c, ldx, dtype, wbuf_dev, stats_ring_dev, xdim, ydim, batch_steps,
                                           g_begin, g_end, total_steps, a0, a1, r0, r1, workspace_dev, workspace_bytes,
                                           flags, nullptr, stream);
}

// equal steps (the round-1/2 entry points): total_steps = num_passes * batch_steps
PXSOM_EXPORT int pxsom_batch_train_steps_sharded(const void *x_dev, int64_t n, int c, int64_t ldx, int dtype,
                                                 double *wbuf_dev, double *stats_ring_dev, int xdim, int ydim,
                                                 int batch_steps, int g_begin, int g_end, int total_steps, double a0,
                                                 double a1, double r0, double r1, void *workspace_dev,
                                                 size_t workspace_bytes, int flags, pxsom_comm *comm, void *stream)
{
    if (batch_steps < 1 || batch_steps > PXSOM_MAX_SCHED_STEPS || total_steps < 1 || total_steps % batch_steps != 0)
        return pxsom::fail(PXSOM_ERR_INVALID_ARG, "pxsom_batch_train_steps: steps [%d, %d) of %d, %d per pass (1..%d, whole passes)",
                           g_begin, g_end, total_steps, batch_steps, PXSOM_MAX_SCHED_STEPS);
    const std::vector<int32_t> e = equal_edges(batch_steps);
    return pxsom_batch_train_sched(x_dev, n, c, ldx, dtype, wbuf_dev, stats_ring_dev, xdim, ydim, batch_steps, e.data(),
                                   batch_steps, g_begin, g_end, total_steps / batch_steps, a0, a1, r0, r1, 0.0, workspace_dev,
                                   workspace_bytes, flags, comm, stream);
}

namespace {
int finish_at(const double *wbuf_dev, const double *stats_ring_dev, int xdim, int ydim, int c, int steps_done, int64_t pos,
              int64_t span, double a0, double a1, double r0, double r1, double *w_out_dev, void *stream)
{
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int k = xdim * ydim, g = steps_done - 1;
    const size_t nw = (size_t)k * c, nstats = (size_t)k * (c + 1);
    const double *w_last = wbuf_dev + (size_t)(g % 2) * nw, *s_last = stats_ring_dev + (size_t)(g % 3) * nstats;
    double thr, alpha;
    batch_schedule(pos, span, a0, a1, r0, r1, &thr, &alpha);
    // one launch: the update reads W of the last step where it lies and writes the result to w_out (no copy in front)
    const size_t stage_bytes = (size_t)k * (c + 1) * sizeof(double);
    const int stage = stage_bytes <= 60 * 1024;
    hipLaunchKernelGGL(batch_update_kernel, dim3(k), dim3(256), stage ? stage_bytes : 0, st, w_out_dev, xdim, ydim, c, s_last,
                       s_last + nw, thr, 1.0 - alpha, pxsom_bmu::batch_gain_saturation(1.0 - alpha), stage, (double *)nullptr, 0, w_last);
    PXSOM_LAUNCH_CHECK("batch_update_kernel");
    return PXSOM_OK;
}
}  // namespace

PXSOM_EXPORT int pxsom_batch_train_sched_finish(const double *wbuf_dev, const double *stats_ring_dev, int xdim, int ydim,
                                                int c, int phases, const int32_t *edges, int steps_per_pass, int steps_done,
                                                int num_passes, double a0, double a1, double r0, double r1,
                                                double *w_out_dev, void *stream)
{
    if (xdim < 1 || ydim < 1 || (int64_t)xdim * ydim > PXSOM_MAX_NODES || c < 1 || c > PXSOM_MAX_CHANNELS)
        return pxsom::fail(PXSOM_ERR_UNSUPPORTED, "pxsom_batch_train_finish: shape %dx%d x %d", xdim, ydim, c);
    int rc = check_sched("pxsom_batch_train_finish", phases, edges, steps_per_pass);
    if (rc) return rc;
    if (!wbuf_dev || !stats_ring_dev || !w_out_dev || num_passes < 1 || steps_done < 1 ||
        (int64_t)steps_done > (int64_t)num_passes * steps_per_pass)
        return pxsom::fail(PXSOM_ERR_INVALID_ARG, "pxsom_batch_train_finish: bad arguments");
    const Sched sc{phases, steps_per_pass, edges};
    return finish_at(wbuf_dev, stats_ring_dev, xdim, ydim, c, steps_done, sc.pos(steps_done - 1), (int64_t)num_passes * phases,
                     a0, a1, r0, r1, w_out_dev, stream);
}

// equal steps: the position of step g of total_steps is g / total_steps whatever the steps per pass
PXSOM_EXPORT int pxsom_batch_train_finish(const double *wbuf_dev, const double *stats_ring_dev, int xdim, int ydim, int c,
                                          int steps_done, int total_steps, double a0, double a1, double r0, double r1,
                                          double *w_out_dev, void *stream)
{
    if (xdim < 1 || ydim < 1 || (int64_t)xdim * ydim > PXSOM_MAX_NODES || c < 1 || c > PXSOM_MAX_CHANNELS)
        return pxsom::fail(PXSOM_ERR_UNSUPPORTED, "pxsom_batch_train_finish: shape %dx%d x %d", xdim, ydim, c);
    if (!wbuf_dev || !stats_ring_dev || !w_out_dev || steps_done < 1 || steps_done > total_steps)
        return pxsom::fail(PXSOM_ERR_INVALID_ARG, "pxsom_batch_train_finish: bad arguments");
    return finish_at(wbuf_dev, stats_ring_dev, xdim, ydim, c, steps_done, steps_done - 1, total_steps, a0, a1, r0, r1, w_out_dev,
                     stream);
}

// 1: the steps of this (matrix, shape, schedule) take the one-launch fused kernel; 0: the launch-per-phase route.  A
// multi-rank job agrees on the route before it starts (MIN over the ranks; PXSOM_TRAIN_UNFUSED for everyone otherwise):
// the two routes produce the same statistics but round the codebook's last bits differently.
PXSOM_EXPORT int pxsom_batch_train_fused_route(const void *x_dev, int c, int64_t ldx, int dtype, int xdim, int ydim, int phases)
{
    if (!pxsom::dtype_ok(dtype) || phases < 1) return 0;
    const int64_t gs = (int64_t)phases * ldx;
    if (dtype == PXSOM_F32) return pxsom_bmu::step_fused_shape<float>(static_cast<const float *>(x_dev), 1, c, ldx, xdim, ydim, gs);
    if (dtype == PXSOM_F16) return pxsom_bmu::step_fused_shape<_Float16>(static_cast<const _Float16 *>(x_dev), 1, c, ldx, xdim, ydim, gs);
    return pxsom_bmu::step_fused_shape<double>(static_cast<const double *>(x_dev), 1, c, ldx, xdim, ydim, gs);
}

// ---- reproducible statistics for binary64 rows (include/pxsom.h) ----------------------------------------------------
PXSOM_EXPORT double pxsom_exact_sum_quantum(double value_bound, int64_t rows_bound)
{
    if (!(value_bound > 0.0) || !(value_bound <= DBL_MAX)) return 0.0;   // all-zero / unbounded data: nothing to round to
    if (rows_bound < 2) rows_bound = 2;
    // sums stay below rows * bound < 2^e; with q = 2^(e - 52) they are multiples of q below 2^52 q: exactly representable,
    // and so is every partial sum in any order
    int e = 0;
    frexp(value_bound, &e);                    // value_bound < 2^e
    int r = 0;
    while (((int64_t)1 << r) < rows_bound && r < 62) r++;
    return ldexp(1.0, e + r - 52);
}

namespace {
template <typename T>
__global__ __launch_bounds__(256) void absmax_kernel(const T *__restrict__ x, int64_t n, int c, int64_t ldx,
                                                     unsigned long long *out)
{
    double m = 0.0;
    const int64_t total = n * c;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t row = e / c;
        const double v = fabs((double)x[row * ldx + (e - row * c)]);
        if (v <= DBL_MAX && v > m) m = v;      // (NaN and Inf fail the first test)
    }
    m = -pxsom::wave_min_f64(-m);
    // non-negative binary64 numbers order like their bit patterns
    if ((threadIdx.x & 63) == 0 && m > 0.0) atomicMax(out, (unsigned long long)__double_as_longlong(m));
}
}  // namespace

PXSOM_EXPORT int pxsom_absmax(const void *x_dev, int64_t n, int c, int64_t ldx, int dtype, double *out_dev, void *stream)
{
    int rc = check_matrix("pxsom_absmax", x_dev, n, c, ldx, dtype);
    if (rc) return rc;
    if (!out_dev) return pxsom::fail(PXSOM_ERR_INVALID_ARG, "pxsom_absmax: null output");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    PXSOM_HIP_TRY(hipMemsetAsync(out_dev, 0, sizeof(double), st));
    if (n == 0) return PXSOM_OK;
    const int64_t grid = std::min<int64_t>((n * c + 255) / 256, (int64_t)pxsom::device_cu_count() * 8);
    unsigned long long *out = reinterpret_cast<unsigned long long *>(out_dev);
    if (dtype == PXSOM_F32)
        hipLaunchKernelGGL(absmax_kernel<float>, dim3((unsigned)grid), dim3(256), 0, st, static_cast<const float *>(x_dev), n, c, ldx, out);
    else if (dtype == PXSOM_F16)
        hipLaunchKernelGGL(absmax_kernel<_Float16>, dim3((unsigned)grid), dim3(256), 0, st, static_cast<const _Float16 *>(x_dev), n, c, ldx, out);
    else
        hipLaunchKernelGGL(absmax_kernel<double>, dim3((unsigned)grid), dim3(256), 0, st, static_cast<const double *>(x_dev), n, c, ldx, out);
    PXSOM_LAUNCH_CHECK("absmax_kernel");
    return PXSOM_OK;
}
