// pxsom_prep.h -- prep_body: everything the BMU filter needs to know about a codebook (fp16 hi/lo MFMA
// fragments, bias = -|w|^2/2, power-of-two scale, error-bound constants, duplicate-node mask), computed by one
// workgroup from a codebook that is already where `wl` points (LDS when it fits).  Used by bmu_prep_kernel
// (pxsom_assign.hip: results go to the assign workspace in HBM) and by the accumulating filter
// (pxsom_assign_filter_acc.hip: every workgroup prepares for itself, results stay in LDS).
#pragma once
#include <cfloat>
#include <cmath>

#include "pxsom_assign.h"
#include "pxsom_wave.h"

namespace pxsom_bmu {
namespace {

// scripts/ubench/assign_phase_timing.hip includes this file with PXSOM_PHASE_TIMING defined: s_memtime at
// phase boundaries of the single-workgroup prep kernel and of workgroup 0 of the exact kernel
#ifdef PXSOM_PHASE_TIMING
__device__ long long g_phase_ticks[32];
#define PXSOM_PHASE(i)                                          \
    do {                                                        \
        if (threadIdx.x == 0 && blockIdx.x == 0) g_phase_ticks[i] = clock64(); \
    } while (0)
#ifdef PXSOM_PHASE_BLOCK0_ONLY
#define PXSOM_PHASE_ANY(i) PXSOM_PHASE(i)
#else
#define PXSOM_PHASE_ANY(i)                                  \
    do {                                                    \
        if (threadIdx.x == 0) g_phase_ticks[i] = clock64(); \
    } while (0)
#endif
#else
#define PXSOM_PHASE(i) \
    do {               \
    } while (0)
#define PXSOM_PHASE_ANY(i) \
    do {                   \
    } while (0)
#endif

// ------------------------------------------------------------------------------------------------
// 1. prep: one workgroup of NT threads.  prep_body works on a codebook that is already in `wl` (LDS when
//    it fits, else HBM); the callers differ in how it got there.
// ------------------------------------------------------------------------------------------------
// MAXK: the largest k the caller can have (sizes the static per-node scratch: 21 bytes per node -- the accumulating
// filter's workgroups, k <= 128, must not carry the 22 KB that 1024 nodes need, or only one of them fits a CU).
template <int NT, int MAXK = PXSOM_MAX_NODES>
__device__ __forceinline__ void prep_body(const double *wl, int k, int c, AssignHdr *hdr, half8 *wfrag,
                                          f32x4 *bias, int nb, int nch, int cpl, int idx_bits, int node_bits,
                                          double *wt_out = nullptr, float *w32_out = nullptr, int cp32 = 0, int npk = 0,
                                          bool center = false)
{
    // binary32 copy, rows zero-padded to cp32 channels: what the long-list exact kernel screens with
    if (w32_out) {
        int node = (int)threadIdx.x / cp32, j = (int)threadIdx.x - node * cp32;
        const int dn = NT / cp32, dj = NT % cp32;
#pragma unroll 4
        for (int e = threadIdx.x; e < k * cp32; e += NT) {
            w32_out[e] = j < c ? (float)wl[(size_t)node * c + j] : 0.f;
            node += dn;
            j += dj;
            if (j >= cp32) {
                j -= cp32;
                node++;
            }
        }
    }
    // big codebooks: a transposed copy [c][k] for the exact kernel (coalesced reads, lanes <-> nodes)
    // Written in ITS order (consecutive threads <-> consecutive nodes of a channel: coalesced stores, strided
    // reads of the staged codebook), the (channel, node) pair advanced without a division per element.
    if (wt_out) {
        int j = (int)threadIdx.x / k, node = (int)threadIdx.x - j * k;
        const int dj = NT / k, dn = NT % k;
#pragma unroll 4
        for (int e = threadIdx.x; e < k * c; e += NT) {
            wt_out[e] = wl[(size_t)node * c + j];
            j += dj;
            node += dn;
            if (node >= k) {
                node -= k;
                j++;
            }
        }
    }
    __shared__ double s_norm2[MAXK];
    __shared__ unsigned long long s_key[MAXK];  // hash of the row's bit patterns (duplicate test)
    __shared__ double s_red[3 * (NT / 64)];
    __shared__ double s_mud[kFilterMaxChannels];   // centring vector (unscaled; binary32-representable values): zeros when off
    __shared__ int s_bad;
    const int tid = threadIdx.x;
    if (tid == 0) s_bad = 0;
    for (int j = tid; j < kFilterMaxChannels; j += NT) s_mud[j] = 0.0;
    __syncthreads();
    // centring (every filter but the one on binary16 rows, c <= 128): mu = the nodes' mean per channel, rounded to binary32 --
    // any vector would do for the ranking, the mean keeps the centred norms (what the error bound is relative to) small.
    // NT / 128 adjacent lanes share a channel.
    if (center) {
        // (a power of two of lanes per channel, whatever NT: workgroups of 768 threads use four, their last 256 threads idle)
        constexpr int per = NT / kFilterMaxChannels >= 8 ? 8 : (NT / kFilterMaxChannels >= 4 ? 4 : (NT / kFilterMaxChannels >= 2 ? 2 : 1));
        const int j = tid / per, part = tid % per;
        double sum = 0.0;
        if (j < c)
            for (int node = part; node < k; node += per) sum += wl[(size_t)node * c + j];
#pragma unroll
        for (int d = 1; d < per; d *= 2) sum += __shfl_xor(sum, d);
        if (part == 0 && j < c) {
            const float m = (float)(sum / (double)k);
            s_mud[j] = fabsf(m) <= 3.0e38f ? (double)m : 0.0;   // (a non-finite codebook lists every row anyway)
        }
        __syncthreads();
    }

    // per-node squared norm (binary64), max |w| and max norm.  `parts` adjacent lanes share a node
    // (interleaved channels, butterfly sum: every node is summed in the same order, so bit-identical
    // rows get bit-identical norms -- the duplicate test below relies on it).
    const int parts = 4 * k <= NT ? 4 : (2 * k <= NT ? 2 : 1);
    const int pshift = parts == 4 ? 2 : (parts == 2 ? 1 : 0);
    double mymax = 0.0, mynorm = 0.0, myraw = 0.0;
    bool bad = false;
    for (int p = tid; p < (k << pshift); p += NT) {
        const int node = p >> pshift, part = p & (parts - 1);
        double sum = 0.0, raw2 = 0.0;
        unsigned long long key = 0;
        for (int j = part; j < c; j += parts) {
            const double vr = wl[(size_t)node * c + j];
            const double v = vr - s_mud[j < kFilterMaxChannels ? j : 0];   // (zeros when the filter is not centred: v == vr, bit for bit)
            bad |= !(fabs(vr) <= DBL_MAX);  // NaN / Inf in the codebook
            sum += v * v;
            raw2 += vr * vr;
            mymax = fmax(mymax, fabs(v));
            const unsigned long long hb = (unsigned long long)__double_as_longlong(vr) * 0x9E3779B97F4A7C15ull +
                                          (unsigned long long)(j + 1) * 0xC2B2AE3D27D4EB4Full;
            key ^= hb ^ (hb >> 29);
        }
        if (parts >= 2) {
            sum += __shfl_xor(sum, 1);
            raw2 += __shfl_xor(raw2, 1);
            key ^= __shfl_xor(key, 1);
        }
        if (parts == 4) {
            sum += __shfl_xor(sum, 2);
            raw2 += __shfl_xor(raw2, 2);
            key ^= __shfl_xor(key, 2);
        }
        if (part == 0) {
            s_norm2[node] = sum;
            s_key[node] = key;
        }
        mynorm = fmax(mynorm, sum);
        myraw = fmax(myraw, raw2);
    }
    if (bad) s_bad = 1;
    // the maxima: DPP wave reduction (max(a, b) = -min(-a, -b)), then 4 partials through LDS
    mymax = -pxsom::wave_min_f64(-mymax);
    mynorm = mynorm == mynorm ? -pxsom::wave_min_f64(-mynorm) : mynorm;
    myraw = myraw == myraw ? -pxsom::wave_min_f64(-myraw) : myraw;
    constexpr int NW = NT / 64;
    if ((tid & 63) == 0) {
        s_red[tid >> 6] = mymax;
        s_red[NW + (tid >> 6)] = mynorm;
        s_red[2 * NW + (tid >> 6)] = myraw;
    }
    __syncthreads();
    PXSOM_PHASE_ANY(2);
    double maxabs = s_red[0], wn2max = s_red[NW], raw2max = s_red[2 * NW];
#pragma unroll
    for (int i = 1; i < NW; i++) {
        maxabs = fmax(maxabs, s_red[i]);
        wn2max = fmax(wn2max, s_red[NW + i]);
        raw2max = fmax(raw2max, s_red[2 * NW + i]);
    }
    // scale = 2^e with maxabs*scale in [128, 256): fp16 keeps 11 significant bits there and the
    // low halves of the split stay normal down to |x| ~ 1e-4 * maxabs.
    int e = 0;
    if (maxabs > 0.0 && maxabs <= DBL_MAX) {
        int ex;
        frexp(maxabs, &ex);  // maxabs = m * 2^ex, m in [0.5, 1)
        e = 8 - ex;
        if (center && raw2max > 0.0 && raw2max <= DBL_MAX) {
            // a codebook that has (nearly) collapsed onto its mean must not blow the scale up: rows lie as far from the
            // mean as they did before, and x' * scale has to stay inside binary16 -- at most 2^6 over what the largest
            // uncentred node norm alone would choose (rows up to ~3.6 |w|max from the mean stay in range)
            int exn;
            frexp(sqrt(raw2max), &exn);
            if (e > 8 - exn + 6) e = 8 - exn + 6;
        }
        if (e > 100) e = 100;
        if (e < -100) e = -100;
    }
    const double scale = ldexp(1.0, e);
    PXSOM_PHASE_ANY(3);
    if (tid == 0) {
        // (maxabs * scale < 256 is what filter_cut_abs counts on: it fails only where the exponent clamp above bit)
        const bool badw = s_bad != 0 || !(wn2max * scale * scale <= 1.0e30) || !(maxabs * scale < 256.0);
        hdr->amb_count = 0;
        hdr->scale = (float)scale;
        // rounded up by a hair; an infinite wn_max makes every row take the exact path
        hdr->wn_max = badw ? 0.f : (float)(sqrt(wn2max) * scale * (1.0 + 1e-6));
        hdr->force_exact = badw ? 1 : 0;  // NaN/Inf/huge codebook: every row takes the exact path
        // coefficient of the rigorous |filter - exact| bound, see DESIGN.md "K7 error bound":
        //   index packing 2^-(23-idx_bits) (idx_bits low mantissa bits replaced), the matrix unit's group additions
        //   filter_accum_units * 2^-24 (the cuts inside its groups: tol_abs), split residual 2^-19,
        //   f64->f32 input rounding 2^-23;  tol = 2 * 1.25 * E
        //   centred filter: + 2^-24, the rounding of x' = fl(x * scale - mu_s) (one fused operation)
        const double coef = ldexp(1.0, -(23 - idx_bits)) +
                            filter_accum_units_for(c, 3, npk) * ldexp(1.0, -24) + ldexp(1.0, -19) + ldexp(1.0, -23) + (center ? ldexp(1.0, -24) : 0.0);
        hdr->tol_rel = (float)(2.5 * coef);
        // first stage of the register-resident filter: Wh*Xh alone.  |X'.W' - Xh.Wh| <= (2^-11 + 2^-11 (1 + 2^-11)) |X'||W'|
        // (binary16 unit roundoff 2^-11 on either factor), C products + the bias in the fp32 accumulation
        const double coef_c = ldexp(1.0, -10) + ldexp(1.0, -20) + ldexp(1.0, -(23 - idx_bits)) + filter_accum_units(c, 1) * ldexp(1.0, -24) +
                              ldexp(1.0, -23) + (center ? ldexp(1.0, -24) : 0.0);
        hdr->tol_rel_coarse = (float)(2.5 * coef_c);
        hdr->tol_abs = (float)filter_tol_abs(c);  // fp16 subnormal floor + the cuts inside the groups of an MFMA (pxsom_assign.h)
        hdr->x_limit = 60000.0f;
        hdr->nb = nb;
        hdr->nch = nch;
        hdr->cpl = cpl;
        hdr->idx_bits = idx_bits;
        hdr->node_bits = node_bits;
        hdr->wn_raw = badw ? 0.f : (float)(sqrt(raw2max) * (1.0 + 1e-6));
        hdr->centred = center ? 1 : 0;
        // a row the filter vouches for has |x' * scale|_2 < x_limit, hence |x_j * scale| < x_limit + max_j |mu_s_j|:
        // below 2^(16 + t) for the smallest such t >= 0
        double mumax = 0.0;
        for (int j = 0; j < (c < kFilterMaxChannels ? c : kFilterMaxChannels); j++) mumax = fmax(mumax, fabs(s_mud[j]) * scale);
        int t = 0;
        while (t < 60 && !(60000.0 + mumax <= ldexp(65536.0, t))) t++;
        hdr->fix_exp = e - t;
    }
    for (int j = tid; j < kFilterMaxChannels; j += NT) hdr->mu_s[j] = (float)(s_mud[j] * scale);

    // A-fragments: wfrag[(b*nsteps + s)*64 + lane], lane = (q<<4 | m): node 16b+m,
    // slot i of lane group q in chunk h <-> channel h*4*cpl + q*cpl + i (i < cpl); s = 2h: hi, 2h+1: lo
    PXSOM_PHASE_ANY(4);
    if (npk > 0) {   // packed K axis (pxsom_assign.h packed_k): fragment m of block b holds slot groups 4m .. 4m + 3
        const int g8 = c / 8;
        for (int f = tid; f < nb * npk * 64; f += NT) {
            const int lane = f & 63, m = (f >> 6) % npk, b = (f >> 6) / npk;
            const int row = lane & 15, q = lane >> 4;
            const int node = node_of_row(b, row, nb);
            const int sg = 4 * m + q, term = sg / g8, ch0 = 8 * (sg - term * g8);
            half8 fr;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                float W = 0.f;
                if (term < 2 && node < k) W = (float)(wl[(size_t)node * c + ch0 + i] * scale);
                const _Float16 hi = (_Float16)W;
                fr[i] = term == 0 ? hi : (_Float16)(W - (float)hi);
            }
            wfrag[(size_t)(b * npk + m) * 64 + lane] = fr;
        }
    }
    const int nsteps = 2 * nch;
    for (int f = tid; f < (npk > 0 ? 0 : nb * nch * 64); f += NT) {
        const int lane = f & 63, h = (f >> 6) % nch, b = (f >> 6) / nch;
        const int m = lane & 15, q = lane >> 4;
        const int node = node_of_row(b, m, nb);
        half8 fhi, flo;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int ch = h * 4 * cpl + q * cpl + i;
            float W = 0.f;
            if (i < cpl && ch < c && node < k) W = (float)((wl[(size_t)node * c + ch] - s_mud[ch < kFilterMaxChannels ? ch : 0]) * scale);
            const _Float16 hi = (_Float16)W;
            fhi[i] = hi;
            flo[i] = (_Float16)(W - (float)hi);
        }
        wfrag[(size_t)(b * nsteps + 2 * h) * 64 + lane] = fhi;
        wfrag[(size_t)(b * nsteps + 2 * h + 1) * 64 + lane] = flo;
    }
    // Exact duplicates of an EARLIER node can never be the answer (their distance is identical and the
    // reference keeps the first minimum), so they are masked out of the filter.  This matters in batch
    // training: while the neighbourhood radius still spans the grid, all central nodes receive the same
    // update and are bit-identical, which would otherwise send every row they win to the exact path.
    // (s_dup reuses s_red's storage class: one flag per node.)
    __shared__ unsigned char s_dup[MAXK];
    PXSOM_PHASE_ANY(5);
    for (int node = tid; node < k; node += NT) s_dup[node] = 0;
    __syncthreads();
    // rows compared 4 channels per trip (the 8 reads of a trip pipeline), leaving at the first trip that
    // differs.  Near-duplicates are the common case in early batch-training steps: gain = 1 makes every
    // node with the same window the same mean up to the last bit, and their norms often round equal.
    auto same_rows = [&](int prev, int node) {
        if (s_key[prev] != s_key[node]) return false;
        const double *a = wl + (size_t)prev * c, *b = wl + (size_t)node * c;
        int j = 0;
        for (; j + 3 < c; j += 4) {
            const bool eq = (a[j] == b[j]) & (a[j + 1] == b[j + 1]) & (a[j + 2] == b[j + 2]) & (a[j + 3] == b[j + 3]);
            if (!eq) return false;
        }
        for (; j < c; j++)
            if (a[j] != b[j]) return false;
        return true;
    };
    {
        // all pairs on the row keys (equal rows have equal keys; norms are useless here: near-duplicates
        // often round to the same norm), the range of earlier nodes split over NT/k threads per node;
        // full channel comparison only on a key match.  (Also for k > 256, where each thread scans for several
        // nodes: a hash table took 60 us at k = 400 -- slot collisions fall back to scans one by one.)
        __shared__ int s_first[MAXK];
        for (int node = tid; node < k; node += NT) s_first[node] = 0x7fffffff;
        __syncthreads();
        int sp = 1;
        while (sp * 2 * k <= NT) sp *= 2;
        const int len = (k + sp - 1) / sp;
        for (int p = tid; p < k * sp; p += NT) {
            const int node = p / sp, part = p - node * sp;
            const int lo = part * len, hi = min(node, lo + len);
            const unsigned long long n2 = s_key[node];
            int hit = 0x7fffffff;
    #pragma unroll 8
            for (int prev = lo; prev < hi; prev++) hit = min(hit, s_key[prev] == n2 ? prev : 0x7fffffff);
            if (hit != 0x7fffffff) atomicMin(&s_first[node], hit);
        }
        __syncthreads();
        for (int node = tid; node < k; node += NT) {
            const int first = s_first[node];
            if (first >= node) continue;
            bool dup = same_rows(first, node);
    #ifdef PXSOM_PHASE_TIMING
            atomicAdd((unsigned long long *)&g_phase_ticks[24], 1ull);
            if (!dup) atomicAdd((unsigned long long *)&g_phase_ticks[25], 1ull);
    #endif
            for (int prev = first + 1; prev < node && !dup; prev++) dup = same_rows(prev, node);  // key collision
            if (dup) s_dup[node] = 1;
        }
    }
    __syncthreads();
    PXSOM_PHASE_ANY(6);
    // bias[b*64 + lane][r] for accumulator row (lane>>4)*4 + r <-> node_of_row(b, 4q + r)
    for (int f = tid; f < nb * 64; f += NT) {
        const int lane = f & 63, b = f >> 6, q = lane >> 4;
        f32x4 bv;
        for (int r = 0; r < 4; r++) {
            const int node = node_of_row(b, q * 4 + r, nb);
            bv[r] = (node < k && !s_dup[node]) ? (float)(-0.5 * s_norm2[node] * scale * scale) : kNegBig;
        }
        bias[f] = bv;
    }
    PXSOM_PHASE_ANY(7);
}

}  // namespace
}  // namespace pxsom_bmu
