// pxsom_batch_step_wide.hip -- ONE launch per BMU-only mini-batch step for codebooks the register-resident step kernel
// (pxsom_batch_step.hip: 10 x 10 grid, C <= 32) cannot hold: any grid of up to 256 nodes, up to 128 channels, binary32 or
// binary64 rows (the cell SOM: 100 nodes x 100 pixel-cluster counts, reference cell_som_clustering.py:8-75; a pixel SOM on a
// grid other than 10 x 10, pixel_som_clustering.py:16-21).
//
// Which steps: those whose pending update has its threshold pinned at 0.5 -- a node's window is the node itself, so the grid's
// shape does not enter -- and that are small enough for their statistics to go to HBM with one device-scope atomic per value and
// workgroup (<= kWideMaxRows rows: every workgroup meets one block of rows).  On the default schedule these are the 16 steps of the tail, each of which otherwise costs four or
// five dependent launches (update + prepare, search, exact, screened exact, sums: ~78 us on config 4 against ~25 us of work).
//
// A workgroup (512 threads) takes 128 rows, each of its eight waves a 16-row tile (round 6; 64 rows on four waves before):
//   P0  requests: the workgroup's rows (HBM: slowest, used last), the counts and sums of step g-1, W_{g-1};
//   P1  the pending update, element-wise and coalesced, in the arithmetic of orc_som_batch_sched (gain = batch_gain(den, 1 - alpha), mean =
//       S (1/den), w + gain (mean - w), no contraction); W_g into LDS (row stride padded to an odd number of words), workgroup 0
//       writes it to HBM;
//   P2  centred norms, maxima, the power-of-two scale, A-fragments (binary16 hi / lo of (W - mu) scale) and biases in LDS;
//   P3  search: the K7 filter on v_mfma_f32_16x16x32_f16 with the three-term split, top-2 with the node id in the low mantissa
//       bits, rigorous tolerance (pxsom_assign_filter_fast.h); a row the filter vouches for has its label, the others wait in a
//       queue;
//   P4  the queue is settled by whichever wave is free, exactly as the oracle does (binary64, j ascending, no contraction, sqrt,
//       first strict minimum) against the LDS copy of W_g: the rest of the labels;
//   P5  the rows join the step's statistics in HBM a RUN OF EQUAL LABELS at a time: labels sorted in LDS, a run's rows summed in
//       registers, one device-scope atomic per (run, channel);
//   and every workgroup clears its share of the next step's statistics buffer.
// Equal nodes need no special case: their scores tie, the row is listed, the exact path takes the first of them.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>

#include "pxsom_assign.h"
#include "pxsom_assign_filter_fast.h"
#include "pxsom_wave.h"

#ifndef PXSOM_WIDE_ABL   // timing builds only (results wrong): 1 no statistics of vouched rows, 2 no listed rows, 4 no search at all,
#define PXSOM_WIDE_ABL 0   // 8 no fragments / biases, 16 no norms
#endif

namespace pxsom_bmu {
namespace {

// Round 6: a workgroup takes 128 rows (eight 16-row tiles) and adds them to the step's statistics ONE VALUE PER RUN OF EQUAL
// LABELS AND CHANNEL, not one per (row, channel): the rows' labels -- the filter's and the exact path's alike -- are sorted by
// label in LDS (a counting sort of 128 small integers), each wave takes an eighth of the sorted rows (lanes <-> channels, all its
// rows requested at once from L2: the tile was just read), sums runs of equal labels in registers and issues a run's atomics
// when the label changes.  With 64 rows per workgroup and one atomic per value a tail step of config 4 (8 333 rows x 100
// columns) issued 842 K device-scope atomics and waited 27 of its 58 us for them (timing builds, profiles/r06/experiments.txt).
constexpr int kWideThreads = 512, kWideWaves = 8, kWideRounds = 1, kWideRowsPerWg = kWideRounds * kWideWaves * 16;
#ifndef PXSOM_WIDE_MAX_ROWS
#define PXSOM_WIDE_MAX_ROWS 65536
#endif
constexpr int64_t kWideMaxRows = PXSOM_WIDE_MAX_ROWS;   // (beyond: a workgroup meets several blocks of rows, each with atomics of its own: the launch-per-phase route's LDS tables win)
constexpr int kWideMaxNodes = 256;

struct WideShape {
    int nb, nch, cpl, cs, kp;   // kp: k rounded up to a multiple of 64 (per-node arrays, node slots of the exact path)
    unsigned cmagic;            // ceil(2^32 / c): e / c == __umulhi(e, cmagic) for e < 2^32 / c (element indices stay below 2^15 c); 0 for c == 1: e itself
    size_t off_frag, off_bias, off_misc, total;
};
inline WideShape wide_shape(int c, int k)
{
    WideShape s;
    s.nb = (k + 15) / 16;
    s.nch = (c + 31) / 32;
    s.cpl = (c + 4 * s.nch - 1) / (4 * s.nch);
    s.cs = (c + 1) | 1;   // W_g row stride, words: odd, and wide enough for [c sums | count] rows while the window sums are formed
    s.kp = (k + 63) & ~63;
    s.cmagic = c > 1 ? (unsigned)((0x100000000ull + (unsigned long long)c - 1ull) / (unsigned long long)c) : 0u;   // (c == 1: 2^32 does not fit)
    s.off_frag = pxsom::align_up((size_t)k * s.cs * sizeof(double), 16);
    s.off_bias = s.off_frag + (size_t)s.nb * 2 * s.nch * 64 * sizeof(half8);
    s.off_misc = s.off_bias + (size_t)s.nb * 64 * sizeof(f32x4);
    // misc: gain[k] | inv[k] | nrm[k] (binary64; after the fragments: the sort's counters, offsets and order) | red[3 * waves] |
    // mu_s[128] (binary32) | labels[rows] (int) | queue[rows] (unsigned short) | control words
    s.total = s.off_misc + (size_t)(3 * s.kp + 3 * kWideWaves) * sizeof(double) + 128 * sizeof(float) +
              kWideRowsPerWg * (sizeof(int) + sizeof(unsigned short)) + 64;
    return s;
}

struct WideCtl {
    unsigned q_n;
    int bad;
    int sorted_n;   // rows with a label in this block
};

template <typename T>
__device__ __forceinline__ double wide_value(T v, double qmagic)
{
    if constexpr (sizeof(T) == 8) return qround((double)v, qmagic);
    else return (double)v;
}

#pragma clang fp contract(off)
// Listed rows settled by a whole wave, TWO at a time (their chains run side by side and share every codebook value read from
// LDS): lanes <-> nodes lane, lane + 64, ... (k <= 256), a row's channels held by the lanes (lane l: channels l and l + 64) and
// broadcast with v_readlane, W_g from LDS (row stride cs, odd: no bank conflicts between the lanes' nodes), distances exactly
// as the oracle forms them (binary64, j ascending, no contraction, sqrt, first strict minimum).  The winners (or -2: no finite
// distance, a NaN row -- not accumulated, as in the oracle) go to *win0 / *win1.
template <typename T, int NS>   // NS: node slots per lane (k <= 64 NS)
__device__ __forceinline__ void wide_exact_rows(const T *xr0, const T *xr1, bool two, int c, int k, const double *wl, int cs, int lane,
                                                int *win0, int *win1)
{
    const double xa0 = lane < c ? (double)xr0[lane] : 0.0, xb0 = lane + 64 < c ? (double)xr0[lane + 64] : 0.0;
    const double xa1 = lane < c ? (double)xr1[lane] : 0.0, xb1 = lane + 64 < c ? (double)xr1[lane + 64] : 0.0;
    const double *wn[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) wn[s] = wl + (size_t)(lane + 64 * s < k ? lane + 64 * s : k - 1) * cs;
    double d0[NS], d1[NS];   // row 0 / row 1, node slot s
#pragma unroll
    for (int s = 0; s < NS; s++) d0[s] = d1[s] = 0.0;
    auto span = [&](int j0, int j1, double va0, double va1, int off) {   // channels [j0, j1), held by lanes j - off
        int j = j0;
        for (; j + 4 <= j1; j += 4) {   // the LDS reads of a trip are issued together; sums stay in j order
            double wv4[NS][4];
#pragma unroll
            for (int s = 0; s < NS; s++)
#pragma unroll
                for (int u = 0; u < 4; u++) wv4[s][u] = wn[s][j + u];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const double x0 = pxsom::readlane_f64(va0, j + u - off), x1 = pxsom::readlane_f64(va1, j + u - off);
#pragma unroll
                for (int s = 0; s < NS; s++) {
                    const double t0 = x0 - wv4[s][u], t1 = x1 - wv4[s][u];
                    d0[s] += t0 * t0;
                    d1[s] += t1 * t1;
                }
            }
        }
        for (; j < j1; j++) {
            const double x0 = pxsom::readlane_f64(va0, j - off), x1 = pxsom::readlane_f64(va1, j - off);
#pragma unroll
            for (int s = 0; s < NS; s++) {
                const double wj = wn[s][j];
                const double t0 = x0 - wj, t1 = x1 - wj;
                d0[s] += t0 * t0;
                d1[s] += t1 * t1;
            }
        }
    };
    span(0, c < 64 ? c : 64, xa0, xa1, 0);
    if (c > 64) span(64, c, xb0, xb1, 64);
    auto settle = [&](const double *d) -> int {
        double best = DBL_MAX;
        int bestk = 0x7fffffff;
#pragma unroll
        for (int s = 0; s < NS; s++) {   // slots in ascending node order: the first strict minimum of this lane's nodes
            const double ds = sqrt(d[s]);
            if (lane + 64 * s < k && ds < best) {
                best = ds;
                bestk = lane + 64 * s;
            }
        }
        const double smin = pxsom::wave_min_f64(best);
        const int win = (int)pxsom::wave_min_u32(best == smin ? (unsigned)bestk : 0xffffffffu);
        return win != 0x7fffffff ? win : -2;
    };
    const int w0 = settle(d0);
    if (lane == 0) *win0 = w0;
    if (two) {
        const int w1 = settle(d1);
        if (lane == 0) *win1 = w1;
    }
}
#pragma clang fp contract(fast)

template <typename T, int NCH>
__global__ __launch_bounds__(kWideThreads) void batch_step_wide_kernel(const T *__restrict__ x, int64_t n, int c, int64_t ldx, int k,
                                                                       double *__restrict__ stats, StepArgs sa, WideShape ws, int xd, int yd,
                                                                       pxsom::RowView rv)
{
    extern __shared__ __attribute__((aligned(16))) char wide_smem[];
    double *wl = reinterpret_cast<double *>(wide_smem);                                   // W_g [k][cs]
    half8 *frag = reinterpret_cast<half8 *>(wide_smem + ws.off_frag);                     // [nb][2 nch][64]
    f32x4 *biasl = reinterpret_cast<f32x4 *>(wide_smem + ws.off_bias);                    // [nb][64]
    double *gain_l = reinterpret_cast<double *>(wide_smem + ws.off_misc);                 // [kp]
    double *inv_l = gain_l + ws.kp, *nrm_l = inv_l + ws.kp;                               // [kp] each
    double *red = nrm_l + ws.kp;                                                          // [3 waves]
    float *mu_s = reinterpret_cast<float *>(red + 3 * kWideWaves);                        // [128]
    int *lab_l = reinterpret_cast<int *>(mu_s + 128);                                     // [rows per workgroup]: node, -1 listed, -2 none
    unsigned short *queue = reinterpret_cast<unsigned short *>(lab_l + kWideRowsPerWg);   // [rows per workgroup]: listed rows (local index)
    WideCtl *ctl = reinterpret_cast<WideCtl *>(queue + kWideRowsPerWg);
    // (the sort of the labels reuses the gains' words: dead once W_g is formed)
    int *hist = reinterpret_cast<int *>(gain_l), *start = hist + ws.kp;                   // [kp] each
    unsigned short *order = reinterpret_cast<unsigned short *>(inv_l);                    // [rows per workgroup] (kp >= 64 binary64 words)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int cs = ws.cs, nb = ws.nb, cpl = ws.cpl;
    const int kc = k * c;
    // ---- P0: this workgroup's first block of rows is requested before anything else (HBM; the search is the last to need them).
    // Lane (q, pix) holds, for chunk h, channels h * 4 cpl + q * cpl + i of row pix of the wave's tile.
    T raw[1][NCH][8];
    auto request = [&](int64_t blk) {
        const int64_t row = blk * kWideRowsPerWg + wv * 16 + (lane & 15);
        const T *xr = x + rv.offset(row < n ? row : (n > 0 ? n - 1 : 0), ldx);
#pragma unroll
        for (int h = 0; h < NCH; h++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int ch = h * 4 * cpl + (lane >> 4) * cpl + i;
                raw[0][h][i] = xr[(i < cpl && ch < c) ? ch : 0];
            }
    };
    constexpr bool kEarly = sizeof(T) < 8;   // (binary64 rows: 64 registers across the update would spill -- requested before the search)
    if constexpr (kEarly)
        if (n > 0 && (int64_t)blockIdx.x * kWideRowsPerWg < n && !(PXSOM_WIDE_ABL & 4)) request(blockIdx.x);

    // ---- P0 / P1: the pending update.  Threshold 0.5: a node's window is the node.  Threshold >= 1 (grids up to 16 x 16, k c <=
    // 16 384): the Chebyshev window sums, separably and in the oracle's order (orc_batch_update: per grid row the window's
    // columns ascending, then the rows ascending) in the W_g region of LDS, node k = x * yd + y as everywhere.
    if (tid == 0) {
        ctl->q_n = 0u;
        ctl->bad = 0;
    }
    const int r = !sa.has_update ? 0 : (sa.thr > 1.0e6 ? 1000000 : (sa.thr >= 1.0 ? (int)floor(sa.thr) : 0));
    if (sa.stats_zero) {
        const int zper = (sa.zero_count + (int)gridDim.x - 1) / (int)gridDim.x;
        const int z1 = min(((int)blockIdx.x + 1) * zper, sa.zero_count);
        for (int e = (int)blockIdx.x * zper + tid; e < z1; e += kWideThreads) sa.stats_zero[e] = 0.0;
    }
    if (r == 0) {
        if (tid < ws.kp) {
            const double den = (tid < k && sa.has_update) ? sa.stats_prev[(size_t)kc + tid] : 0.0;
            gain_l[tid] = den > 0.0 ? batch_gain(den, sa.q, sa.sat) : -1.0;
            inv_l[tid] = den > 0.0 ? 1.0 / den : 0.0;
        }
        __syncthreads();
        {
#pragma clang fp contract(off)
        constexpr int kInFlight = 10;   // elements' loads in flight per thread (100 x 100: two trips instead of five)
        for (int e0 = tid; e0 < kc; e0 += kInFlight * kWideThreads) {
            double wo[kInFlight], sv[kInFlight];
#pragma unroll
            for (int u = 0; u < kInFlight; u++) {
                const int e = e0 + u * kWideThreads < kc ? e0 + u * kWideThreads : 0;
                wo[u] = sa.w_in[e];
                sv[u] = sa.has_update ? sa.stats_prev[e] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < kInFlight; u++) {
                const int e = e0 + u * kWideThreads;
                if (e < kc) {
                    const int node = (ws.cmagic ? (int)__umulhi((unsigned)e, ws.cmagic) : e), j = e - node * c;
                    const double gain = gain_l[node];
                    double v = wo[u];
                    if (gain >= 0.0) {
                        const double mean = sv[u] * inv_l[node];
                        v = gain == 1.0 ? mean : v + gain * (mean - v);
                    }
                    wl[(size_t)node * cs + j] = v;
                    if (sa.w_out && blockIdx.x == 0) sa.w_out[e] = v;
                }
            }
        }
        }
    } else {
#pragma clang fp contract(off)
        constexpr int kMaxE = 32, kMaxD = 16;           // elements of W per thread; grid side (checked by the host)
        const int nc = c + 1;
        double wold[kMaxE];
#pragma unroll
        for (int u = 0; u < kMaxE; u++) wold[u] = sa.w_in[tid + kWideThreads * u < kc ? tid + kWideThreads * u : 0];
        for (int e = tid; e < k * nc; e += kWideThreads) {   // statistics rows [c sums | count] into LDS
            const int node = e / nc, cc = e - node * nc;
            wl[(size_t)node * cs + cc] = cc < c ? sa.stats_prev[(size_t)node * c + cc] : sa.stats_prev[(size_t)kc + node];
        }
        __syncthreads();
        // along y: line (x, cc); then along x: line (y, cc).  A line is read into registers, its window sums written back.
        auto pass = [&](int nlines_major, int len, int stride, bool along_y) {
            for (int line = tid; line < nlines_major * nc; line += kWideThreads) {
                const int major = line / nc, cc = line - major * nc;
                double *base = wl + (along_y ? (size_t)major * yd * cs : (size_t)major * cs) + cc;
                double v[kMaxD];
#pragma unroll
                for (int i = 0; i < kMaxD; i++) v[i] = i < len ? base[(size_t)i * stride] : 0.0;
#pragma unroll
                for (int p = 0; p < kMaxD; p++) {
                    if (p < len) {
                        double t = 0.0;
#pragma unroll
                        for (int i = 0; i < kMaxD; i++)
                            if (i < len && i >= p - r && i <= p + r) t += v[i];
                        base[(size_t)p * stride] = t;
                    }
                }
            }
        };
        pass(xd, yd, cs, true);
        __syncthreads();
        pass(yd, xd, yd * cs, false);
        __syncthreads();
        if (tid < ws.kp) {
            const double den = tid < k ? wl[(size_t)tid * cs + c] : 0.0;
            gain_l[tid] = den > 0.0 ? batch_gain(den, sa.q, sa.sat) : -1.0;
            inv_l[tid] = den > 0.0 ? 1.0 / den : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < kMaxE; u++) {   // new node values into registers: every read of the window sums comes first
            const int e = tid + kWideThreads * u;
            if (e < kc) {
                const int node = (ws.cmagic ? (int)__umulhi((unsigned)e, ws.cmagic) : e), j = e - node * c;
                const double gain = gain_l[node];
                if (gain >= 0.0) {
                    const double mean = wl[(size_t)node * cs + j] * inv_l[node];
                    wold[u] = gain == 1.0 ? mean : wold[u] + gain * (mean - wold[u]);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < kMaxE; u++) {
            const int e = tid + kWideThreads * u;
            if (e < kc) {
                const int node = (ws.cmagic ? (int)__umulhi((unsigned)e, ws.cmagic) : e), j = e - node * c;
                wl[(size_t)node * cs + j] = wold[u];
                if (sa.w_out && blockIdx.x == 0) sa.w_out[e] = wold[u];
            }
        }
    }
    if (tid < 128) mu_s[tid] = (sa.mu32 && tid < c) ? sa.mu32[tid] : 0.f;   // (unscaled until the scale is known)
    const float mu_norm = sa.mu32 ? sa.mu32[kFilterMaxChannels] : 0.f;
    __syncthreads();

    // ---- P2: centred norms, maxima, scale; fragments and biases
    {
        const int part = tid & 3;
        double mymax = 0.0, mynrm = 0.0;
        bool bad = false;
        for (int node = tid >> 2; node < ws.kp && !(PXSOM_WIDE_ABL & 16); node += kWideThreads / 4) {   // (uniform trip count: the shuffles need every lane)
            double nrm = 0.0;
            if (node < k) {
                for (int j = part; j < c; j += 4) {
                    const double v = wl[(size_t)node * cs + j];
                    bad |= !(fabs(v) <= DBL_MAX);
                    const double vc = v - (double)mu_s[j];
                    nrm += vc * vc;
                    mymax = fmax(mymax, fabs(vc));
                }
            }
            nrm += __shfl_xor(nrm, 1);
            nrm += __shfl_xor(nrm, 2);
            if (node < k && part == 0) nrm_l[node] = nrm;
            if (node < k && nrm == nrm) mynrm = fmax(mynrm, nrm);
        }
        if (bad) ctl->bad = 1;
        const double wmax = -pxsom::wave_min_f64(-(mymax == mymax ? mymax : 0.0));
        const double nmax = -pxsom::wave_min_f64(-mynrm);
        if (lane == 0) {
            red[wv] = wmax;
            red[kWideWaves + wv] = nmax;
        }
    }
    __syncthreads();
    double maxabs = red[0], wn2max = red[kWideWaves];
#pragma unroll
    for (int i = 1; i < kWideWaves; i++) {
        maxabs = fmax(maxabs, red[i]);
        wn2max = fmax(wn2max, red[kWideWaves + i]);
    }
    int sexp = 0;
    if (maxabs > 0.0 && maxabs <= DBL_MAX) {
        int ex;
        frexp(maxabs, &ex);
        sexp = 8 - ex;
        if (mu_norm > 0.f) {   // a codebook collapsed onto the centring vector must not blow the scale up (pxsom_batch_step.hip P4)
            int exn;
            frexpf(mu_norm, &exn);
            if (sexp > 8 - exn + 6) sexp = 8 - exn + 6;
        }
        sexp = max(-100, min(100, sexp));
    }
    const double scale_d = ldexp(1.0, sexp);
    const bool badw = ctl->bad != 0 || !(wn2max * scale_d * scale_d <= 1.0e30) || !(maxabs * scale_d < 256.0);   // (256: filter_cut_abs)
    const float scale = (float)scale_d;
    const float wn_max = badw ? 0.f : (float)(sqrt(wn2max) * scale_d * (1.0 + 1e-6));
    const bool force_exact = badw;
    const float x_limit = 60000.0f;
    // A-fragments: frag[(b * 2 nch + 2 h) * 64 + lane] hi, + 1: lo; lane (q << 4 | m) <-> node 16 b + m, slot i of lane group q in
    // chunk h <-> channel h * 4 cpl + q * cpl + i
    for (int f = tid; f < nb * NCH * 64 && !(PXSOM_WIDE_ABL & 8); f += kWideThreads) {
        const int fl = f & 63, h = (f >> 6) % NCH, b = (f >> 6) / NCH;
        const int m = fl & 15, q = fl >> 4, node = 16 * b + m;
        half8 fhi, flo;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int ch = h * 4 * cpl + q * cpl + i;
            float W = 0.f;
            if (i < cpl && ch < c && node < k) W = (float)((wl[(size_t)node * cs + ch] - (double)mu_s[ch]) * scale_d);
            const _Float16 hi = (_Float16)W;
            fhi[i] = hi;
            flo[i] = (_Float16)(W - (float)hi);
        }
        frag[(size_t)(b * 2 * NCH + 2 * h) * 64 + fl] = fhi;
        frag[(size_t)(b * 2 * NCH + 2 * h + 1) * 64 + fl] = flo;
    }
    for (int f = tid; f < nb * 64; f += kWideThreads) {
        const int fl = f & 63, b = f >> 6, q = fl >> 4;
        f32x4 bv;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int node = 16 * b + 4 * q + r;
            bv[r] = node < k ? (float)(-0.5 * nrm_l[node] * scale_d * scale_d) : kNegBig;
        }
        biasl[(size_t)b * 64 + fl] = bv;
    }
    __syncthreads();   // (everybody has read the unscaled centring vector)
    if (tid < 128) mu_s[tid] = (float)((double)mu_s[tid] * scale_d);
    __syncthreads();

    // ---- P3: search
    const int pix = lane & 15, q = lane >> 4;
    const unsigned idx_mask = nb > 8 ? 63u : 31u;   // (b * 4 + r): the lane group travels beside the score, not inside it
    for (int64_t blk = blockIdx.x; blk * kWideRowsPerWg < n && !(PXSOM_WIDE_ABL & 4); blk += gridDim.x) {
        const int64_t row0 = blk * kWideRowsPerWg;
        if constexpr (!kEarly) request(blk);
#pragma unroll
        for (int rd = 0; rd < kWideRounds; rd++) {
            const int rl = (rd * kWideWaves + wv) * 16 + pix;   // the row's place in the workgroup's block
            const bool valid = row0 + rl < n;
            constexpr int slot = 0;
            half8 bh[NCH], bl[NCH];
            float ss = 0.f;
#pragma unroll
            for (int h = 0; h < NCH; h++)
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int ch = h * 4 * cpl + q * cpl + i;
                    float xs = 0.f;
                    if (i < cpl && ch < c) {
                        if constexpr (sizeof(T) == 8) xs = (float)__builtin_fma((double)raw[slot][h][i], scale_d, -(double)mu_s[ch]);
                        else xs = fmaf((float)raw[slot][h][i], scale, -mu_s[ch]);
                    }
                    const _Float16 hi = (_Float16)xs;
                    bh[h][i] = hi;
                    bl[h][i] = (_Float16)(xs - (float)hi);
                    ss = fmaf((float)hi, (float)hi, ss);
                }
            float m1 = kNegBig, m2 = kNegBig;
            for (int b = 0; b < nb; b++) {
                // (the cross terms in an accumulator of their own, joined at the end: filter_accum_units_split, pxsom_assign.h)
                f32x4 acc = biasl[(size_t)b * 64 + lane], accx = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int h = 0; h < NCH; h++) {
                    const half8 whi = frag[(size_t)(b * 2 * NCH + 2 * h) * 64 + lane], wlo = frag[(size_t)(b * 2 * NCH + 2 * h + 1) * 64 + lane];
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi, bh[h], acc, 0, 0, 0);
                    accx = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi, bl[h], accx, 0, 0, 0);
                    accx = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo, bh[h], accx, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; r++) acc[r] = acc[r] + accx[r];
                top2_quad(m1, m2, pack_idx(acc[0], (unsigned)(b * 4 + 0), idx_mask), pack_idx(acc[1], (unsigned)(b * 4 + 1), idx_mask),
                          pack_idx(acc[2], (unsigned)(b * 4 + 2), idx_mask), pack_idx(acc[3], (unsigned)(b * 4 + 3), idx_mask));
            }
            // merge the four lane groups of a pixel (every lane of the pixel ends up with the result); wq: the group the best
            // score came from (on equal scores either: such a row is listed)
            float f1 = m1, f2 = m2;
            unsigned wq = (unsigned)q;
            {
                const F2 e1 = xchg16(f1), e2 = xchg16(f2), e3 = xchg16(ss);
                const float best = fmaxf(e1.a, e1.b);
                if (best != f1) wq ^= 1u;      // the partner group's score won (q ^ 1: neither has merged before)
                f1 = best;
                f2 = fmaxf(fmaxf(fminf(e1.a, e1.b), e2.a), e2.b);
                ss = e3.a + e3.b;
            }
            {
                const F2 e1 = xchg32(f1), e2 = xchg32(f2), e3 = xchg32(ss), eq = xchg32(__uint_as_float(wq));
                const float best = fmaxf(e1.a, e1.b);
                const unsigned qa = __float_as_uint(eq.a), qb = __float_as_uint(eq.b);
                if (best != f1) wq = qa == wq ? qb : qa;   // the other half's winner
                f1 = best;
                f2 = fmaxf(fmaxf(fminf(e1.a, e1.b), e2.a), e2.b);
                ss = e3.a + e3.b;
            }
            const float xn = __builtin_amdgcn_sqrtf(ss) * 1.001f;
            const bool finite_n = (__float_as_uint(ss) & 0x7f800000u) != 0x7f800000u;
            const float tol = sa.tol_rel * (xn * wn_max + 0.5f * wn_max * wn_max) + sa.tol_abs * (xn + wn_max) + kTolFloor;
            const bool amb = !((f1 - f2) > tol) || !(xn < x_limit) || !finite_n || force_exact;
            const unsigned id = __float_as_uint(f1) & idx_mask;
            const int node = (int)(16u * (id >> 2) + 4u * wq + (id & 3u));
            // a listed row waits in the queue; every row's label ends up in lab_l (-2: past the end, or a row without a finite distance)
            if (q == 0) lab_l[rl] = !valid ? -2 : (amb ? -1 : node);
            if (valid && amb && q == 0) queue[atomicAdd(&ctl->q_n, 1u)] = (unsigned short)rl;
        }
        if constexpr (kEarly)
            if ((blk + gridDim.x) * kWideRowsPerWg < n) request(blk + gridDim.x);   // (the next block's rows, if this workgroup has one)
        // ---- P4: the listed rows of this block, two at a time, by whichever wave is free: their labels join the others'
        __syncthreads();
        const unsigned queued = (PXSOM_WIDE_ABL & 2) ? 0u : ctl->q_n;
        for (unsigned i = 2 * wv; i < queued; i += 2 * kWideWaves) {
            const bool two = i + 1 < queued;
            const int ra = queue[i], rb = queue[two ? i + 1 : i];
            const T *xa = x + rv.offset(row0 + ra, ldx), *xb = x + rv.offset(row0 + rb, ldx);
            if (k <= 128) wide_exact_rows<T, 2>(xa, xb, two, c, k, wl, cs, lane, lab_l + ra, lab_l + rb);
            else wide_exact_rows<T, 4>(xa, xb, two, c, k, wl, cs, lane, lab_l + ra, lab_l + rb);
        }
        for (int i = tid; i < ws.kp; i += kWideThreads) hist[i] = 0;
        __syncthreads();
        // ---- P5: the block's rows into the statistics, a label at a time.  Counting sort of the labels (place within a label: the
        // order the counter's atomics were served in -- the sums are exact, so any order gives the same bits), then wave w takes
        // the labels w, w + 8, ...: lanes <-> channels l and l + 64, the label's rows eight loads at a time, one atomic per value.
        int mylab = -1, mypos = 0;
        if (tid < kWideRowsPerWg) {
            mylab = lab_l[tid];
            if (mylab >= 0) mypos = atomicAdd(&hist[mylab], 1);
        }
        __syncthreads();
        if (wv == 0) {   // exclusive prefix sums of the counters: lane l holds the nodes l * NS .. l * NS + NS - 1 (kp = 64 NS <= 256)
            const int ns = ws.kp >> 6;
            int mine[4], tot = 0;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                mine[u] = u < ns ? hist[lane * ns + u] : 0;
                tot += mine[u];
            }
            int incl = tot;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int up = __shfl_up(incl, d);
                if (lane >= d) incl += up;
            }
            int run = incl - tot;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (u < ns) start[lane * ns + u] = run;
                run += mine[u];
            }
            if (lane == 63) ctl->sorted_n = run;
        }
        __syncthreads();
        if (mylab >= 0) order[start[mylab] + mypos] = (unsigned short)tid;
        __syncthreads();
        if (!(PXSOM_WIDE_ABL & 1)) {
            constexpr int kPer = kWideRowsPerWg / kWideWaves;   // sorted rows a wave takes at most
            const int m = ctl->sorted_n, per = (m + kWideWaves - 1) / kWideWaves;
            const int p0 = wv * per, p1 = min(m, p0 + per);
            T va[kPer], vb[kPer];
            int labs[kPer];
#pragma unroll
            for (int u = 0; u < kPer; u++) {
                labs[u] = -1;
                va[u] = vb[u] = (T)0;
                if (p0 + u < p1) {
                    const int r = order[p0 + u];
                    labs[u] = lab_l[r];
                    const T *xq = x + rv.offset(row0 + r, ldx);
                    va[u] = xq[lane < c ? lane : 0];
                    vb[u] = xq[lane + 64 < c ? lane + 64 : 0];
                }
            }
            auto emit = [&](int lab, double s0, double s1, int cnt) {
                double *dst = stats + (size_t)lab * c;
                if (lane < c) __hip_atomic_fetch_add(dst + lane, s0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (lane + 64 < c) __hip_atomic_fetch_add(dst + lane + 64, s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (lane == 0) __hip_atomic_fetch_add(stats + (size_t)kc + lab, (double)cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            };
            int cur = -1, cnt = 0;
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int u = 0; u < kPer; u++) {
                if (labs[u] >= 0) {   // (wave-uniform)
                    if (labs[u] != cur) {
                        if (cur >= 0) emit(cur, s0, s1, cnt);
                        cur = labs[u];
                        s0 = s1 = 0.0;
                        cnt = 0;
                    }
                    s0 += wide_value<T>(va[u], sa.qmagic);
                    s1 += wide_value<T>(vb[u], sa.qmagic);
                    cnt++;
                }
            }
            if (cur >= 0) emit(cur, s0, s1, cnt);
        }
        __syncthreads();
        if (tid == 0) ctl->q_n = 0u;
        __syncthreads();
    }
}

}  // namespace

// shapes the kernel covers (a codebook whose LDS copy + fragments fit one CU)
template <typename T>
bool step_wide_shape(int c, int k)
{
    if (sizeof(T) < 4 || c < 1 || c > kFilterMaxChannels || k < 1 || k > kWideMaxNodes) return false;
    return wide_shape(c, k).total <= 150 * 1024;
}
template bool step_wide_shape<float>(int, int);
template bool step_wide_shape<double>(int, int);
template bool step_wide_shape<_Float16>(int, int);

int64_t step_wide_max_rows() { return kWideMaxRows; }
// steps with a window (threshold >= 1) and the first step of a run (no pending update): the grid's sides and the codebook's
// size are bounded by what a thread keeps in registers while the window sums are formed
bool step_wide_windowed(int xdim, int ydim, int c) { return xdim <= 16 && ydim <= 16 && (int64_t)xdim * ydim * c <= 32 * kWideThreads; }

// One step: the pending update of sa (has_update == 0: none), the search of the n rows x[i * ldx], their statistics added to
// `stats` (cleared by an earlier step), sa.stats_zero cleared, W_g to sa.w_out.
template <typename T>
int launch_batch_step_wide(const T *x, int64_t n, int c, int64_t ldx, int xdim, int ydim, double *stats, const StepArgs &sa, hipStream_t st)
{
    const int k = xdim * ydim;
    const WideShape ws = wide_shape(c, k);
    void (*kern)(const T *, int64_t, int, int64_t, int, double *, StepArgs, WideShape, int, int, pxsom::RowView) = nullptr;
    switch (ws.nch) {
        case 1: kern = batch_step_wide_kernel<T, 1>; break;
        case 2: kern = batch_step_wide_kernel<T, 2>; break;
        case 3: kern = batch_step_wide_kernel<T, 3>; break;
        default: kern = batch_step_wide_kernel<T, 4>; break;
    }
    static pxsom::PerDevice<size_t> raised_on[4];
    size_t &raised = raised_on[ws.nch - 1].here();
    if (raised < ws.total) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ws.total);
        if (e != hipSuccess)
            return pxsom::fail(PXSOM_ERR_HIP, "wide step kernel: cannot raise the LDS limit to %zu bytes: %s", ws.total, hipGetErrorString(e));
        raised = ws.total;
    }
    const int64_t blocks = std::max<int64_t>((n + kWideRowsPerWg - 1) / kWideRowsPerWg, 1);
    const int grid = (int)std::min<int64_t>(blocks, (int64_t)pxsom::device_cu_count());
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kWideThreads), ws.total, st, x, n, c, ldx, k, stats, sa, ws, xdim, ydim, pxsom::current_row_view());
    PXSOM_LAUNCH_CHECK("batch_step_wide_kernel");
    return PXSOM_OK;
}
template int launch_batch_step_wide<float>(const float *, int64_t, int, int64_t, int, int, double *, const StepArgs &, hipStream_t);
template int launch_batch_step_wide<double>(const double *, int64_t, int, int64_t, int, int, double *, const StepArgs &, hipStream_t);

}  // namespace pxsom_bmu
