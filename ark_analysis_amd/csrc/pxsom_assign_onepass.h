// pxsom_assign_onepass.h -- the register-resident BMU search with TWO 16-row tiles per trip (round 5): labels + per-cluster
// fixed-point tables in one pass (pxsom_assign_sums / pxsom_assign_means), or labels alone (pxsom_assign on binary64 rows).
// Same arithmetic contract as bmu_filter_fast (pxsom_assign_filter_fast.h): two-stage search on v_mfma_f32_16x16x32_f16, rows the
// first stage cannot vouch for searched in full from a queue of their wave, rows the full search cannot vouch for settled in
// the oracle's binary64 arithmetic inside the launch, every vouched row added to a 64-bit fixed-point table in LDS.  What
// differs is the shape:
//   * two tiles per trip instead of four: row sets, fp16 operands, accumulators and the top-2 state halve;
//   * C <= 24: the bias -|W'|^2 / 2 rides in the MFMA's spare k-slots 6, 7 of every lane group -- three binary16 parts of the
//     bias against constant multipliers on the row side, the accumulator starts from the inline constant 0 and the 28 bias
//     registers are gone; a fourth spare slot masks duplicate nodes (-65504 x 65504).  C = 26..32: bias registers as before;
//   * with two tiles the transposing merge leaves tile 0's rows in the even lane rows and tile 1's in the odd ones, twice over:
//     the labels a lane needs for its table adds come from ONE v_permlane16_swap (no LDS round trip behind the atomics);
//   * self-contained: every workgroup prepares the codebook itself, no workspace, no list, no launch behind it.
// Where it is used (pxsom_assign_filter_acc.hip, pxsom_assign.hip) and what was measured (profiles/r05/):
//   * BINARY64 ROWS -- what the drop-in classes hold -- by default (512 threads, 256 VGPRs, no spills: bmu_filter_fast keeps four
//     tiles of binary64 rows in flight and spills 28 - 138 VGPRs).  4 M x 22 rows: labels + mean table 0.285 -> 0.183 ms,
//     labels 0.220 -> 0.167 ms = 4.4 TB/s of rows (f64_assign_probe.txt); PXSOM_ONEPASS_F64=0: the old kernels (same-box A/B).
//   * binary32 / binary16 rows, C <= 24: opt-in (PXSOM_ONEPASS=1; 768 threads x 1 per CU, three waves per SIMD at 168 VGPRs) --
//     built to test whether a third wave per SIMD lifts the issue-bound one-pass kernel: it does not, 0.297 ms against 0.262
//     for bmu_filter_fast<ACC, FIX> on the bench matrix; the timing builds below (PXSOM_ONE_ABL) put the table adds at 0.08 ms
//     of it, a third of that LDS bank conflicts (onepass_three_waves_ablation.txt).
#pragma once
#include "pxsom_assign_filter_fast.h"

namespace pxsom_bmu {
namespace {

#ifndef PXSOM_ONE_ABL
#define PXSOM_ONE_ABL 0   // timing builds: 1 = no table adds, 2 = no top-2 bookkeeping, 4 = conflict-free table addresses, 8 = values formed but not added, 16 = half the adds
#endif
#ifndef PXSOM_ONE_THREADS
#define PXSOM_ONE_THREADS 768
#endif
#ifndef PXSOM_ONE_WPE
#define PXSOM_ONE_WPE 3
#endif
constexpr int kOneThreads = PXSOM_ONE_THREADS;   // 768 x 1 per CU: three waves per SIMD (168 VGPRs); 512 x 2: four (128)
constexpr int kOneTiles = 2;          // 16-row tiles per trip
constexpr int kOneNB = 7;             // node blocks (K in (96, 100])
constexpr unsigned kOneS1Queue = 128; // rows a wave can hold back for the full search
constexpr unsigned kOneAmbQueue = 256;

// dynamic LDS of one workgroup (bytes) at k nodes, c channels
__host__ __device__ inline size_t onepass_lds_bytes(int k, int c, int threads)
{
    const int kOneWaves = threads / 64;
    const size_t table = (((size_t)(k + 1) * (acc_stride(c) + 1) + 1) & ~(size_t)1) * sizeof(double);
    return table + (size_t)k * c * sizeof(double) + (size_t)kOneNB * 2 * 64 * sizeof(half8) + (size_t)kOneNB * 64 * sizeof(f32x4) +
           kHdrBytes + kOneAmbQueue * sizeof(int64_t) + 16 + (size_t)kOneWaves * kOneS1Queue * sizeof(int64_t);
}

// THREADS x WPE: 768 x 3 (one workgroup per CU, three waves per SIMD, 168 VGPRs) for binary32 / binary16 rows; 512 x 2 (256 VGPRs)
// for binary64 rows, whose row sets take twice the registers -- for them this kernel is the spill-free one (bmu_filter_fast
// keeps four tiles of binary64 rows in flight: 28 - 138 spilled VGPRs).
// TMODE: 1 = 64-bit fixed-point tables (pxsom_assign_sums / means); 2 = binary64 tables, ds_add_f64 (pxsom_batch_accumulate: the
// batch rule's statistics as its tests pin them); 0 = labels only (pxsom_assign): no table, no flush, listed rows settled for
// their label alone.
template <typename T, int CPL, int THREADS, int WPE, int TMODE = 1>
__global__ __launch_bounds__(THREADS, WPE) void bmu_onepass_kernel(const T *__restrict__ x, int64_t n, int c, int64_t ldx,
                                                                     int32_t *__restrict__ labels, int k,
                                                                     double *__restrict__ stats,
                                                                     const double *__restrict__ wcodes, int idx_bits,
                                                                     int node_bits, int fix_rows_log2)
{
    // FOLD: the bias rides in the spare k-slots 6, 7 of every lane group (C <= 24); C = 26..32 fills the slots: bias registers
    constexpr bool FOLD = CPL <= 6;
    constexpr int kOneThreads = THREADS, kOneWaves = THREADS / 64;
    constexpr int NB = kOneNB, NP = CPL / 2;
    constexpr bool TABLE = TMODE != 0, FIXT = TMODE == 1;
    extern __shared__ __attribute__((aligned(16))) char one_smem[];
    double *ls = reinterpret_cast<double *>(one_smem);
    const int cs = acc_stride(c);
    double *wt = ls + (((size_t)(k + 1) * (cs + 1) + 1) & ~(size_t)1);              // [c][k] transposed binary64 codebook
    half8 *frag_l = reinterpret_cast<half8 *>(wt + (size_t)k * c);                  // [NB][2][64]: hi (bias slots folded in), lo
    f32x4 *bias_l = reinterpret_cast<f32x4 *>(frag_l + NB * 2 * 64);                // [NB][64] (prep_body's output; read by the fold)
    AssignHdr *hdr = reinterpret_cast<AssignHdr *>(bias_l + NB * 64);
    int64_t *amb_q = reinterpret_cast<int64_t *>(reinterpret_cast<char *>(hdr) + kHdrBytes);
    unsigned *amb_n = reinterpret_cast<unsigned *>(amb_q + kOneAmbQueue);
    int64_t *s1_q = reinterpret_cast<int64_t *>(amb_n + 4) + (size_t)(threadIdx.x >> 6) * kOneS1Queue;   // this wave's
    unsigned s1_n = 0u;   // wave-uniform
    bool scan_trip = false;   // wave-uniform: this trip's tiles are summed along the row axis before they touch the table (see the adds)
    const int tid = threadIdx.x;
    {
        // the codebook: row-major copy (in the table's storage, free until the first add) for prep_body, transposed copy for the
        // rows settled exactly
        double *wrow = ls;
        if (tid == 0) *amb_n = 0u;
        int node = tid / c, j = tid - node * c;
        const int dnode = kOneThreads / c, dj = kOneThreads % c;
        for (int e0 = tid; e0 < k * c; e0 += 4 * kOneThreads) {
            double v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = wcodes[e0 + u * kOneThreads < k * c ? e0 + u * kOneThreads : 0];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int e = e0 + u * kOneThreads;
                if (e < k * c) {
                    wrow[e] = v[u];
                    wt[j * k + node] = v[u];
                }
                node += dnode;
                j += dj;
                if (j >= c) {
                    j -= c;
                    node++;
                }
            }
        }
        __syncthreads();
        prep_body<kOneThreads, 128>(wrow, k, c, hdr, frag_l, bias_l, NB, 1, CPL, idx_bits, node_bits, nullptr, nullptr, 0, 0, true);
        __syncthreads();
        // the bias into the spare k-slots of the hi fragments.  A-fragment lane (q, m) of block b holds accumulator row m, whose
        // bias is bias_l[b][lane (m >> 2, *)][m & 3].  bias = 64 a1 + a2 / 32 + a3 / 16384 (three binary16 parts against the
        // constant multipliers 64, 2^-5, 2^-14 on the row side: |bias| <= 2^15 C so a1 stays below 2^14, the parts keep 33
        // bits -- what is lost is below 2^-33 |bias| + 2^-39); a masked node (duplicate of an earlier one, or past k) gets
        // -65504 against the multiplier 65504: -4.29e9, under every score a row inside the filter's range can have
        // (>= -(|X'| + |W'|)^2 / 2 > -1.9e9)
        for (int f = tid; f < (FOLD ? NB * 64 : 0); f += kOneThreads) {
            const int lane = f & 63, b = f >> 6, q = lane >> 4, m = lane & 15;
            const float bv = bias_l[b * 64 + ((m >> 2) << 4)][m & 3];
            _Float16 s6 = (_Float16)0, s7 = (_Float16)0;
            if (bv > 0.5f * kNegBig) {
                const _Float16 a1 = (_Float16)(bv * 0.015625f);
                const float r1 = bv - (float)a1 * 64.f;
                const _Float16 a2 = (_Float16)(r1 * 32.f);
                const float r2 = r1 - (float)a2 * 0.03125f;
                const _Float16 a3 = (_Float16)(r2 * 16384.f);
                if (q == 0) {
                    s6 = a1;
                    s7 = a2;
                } else if (q == 1) {
                    s6 = a3;
                }
            } else if (q == 1) {
                s7 = (_Float16)(-65504.f);
            }
            _Float16 *fr = reinterpret_cast<_Float16 *>(frag_l + (size_t)(b * 2) * 64 + lane);
            fr[6] = s6;
            fr[7] = s7;
        }
        if constexpr (TABLE)
            for (int e = tid; e < (k + 1) * (cs + 1); e += kOneThreads) ls[e] = 0.0;
        __syncthreads();
    }
    constexpr unsigned idx_mask = 127u, node_mask = 127u;
    const float scale = hdr->scale, wn_max = hdr->wn_max, tol_abs = hdr->tol_abs, x_limit = hdr->x_limit;
    // The bias goes through the MFMA's accumulation as three more products in slots 6, 7 of a lane group.  In the unit's measured
    // arithmetic (profiles/r05/mfma_rounding.txt) the eight slots of a group are cut below 2^-24 of the group's LARGEST product --
    // here possibly the bias part |W'|^2 / 2 itself, not max |x' w'| as filter_cut_abs assumes -- so up to seven products of that
    // group lose up to 2^-24 |bias| each, and the group's sum joins the accumulator within 2 x 2^-24: 8 units of the relative term
    // (it is relative to |X'||W'| + |W'|^2 / 2, which bounds |bias|), charged to both tolerances (round 5's advice: 4 were charged).
    const float tol_fold = FOLD ? 2.5f * 8.f * 5.9604645e-8f : 0.f;
    const float tol_rel = hdr->tol_rel + tol_fold, tol_rel_coarse = hdr->tol_rel_coarse + tol_fold;
    const bool force_exact = hdr->force_exact != 0;
    const FixPoint fx = make_fixpoint(hdr->fix_exp, fix_rows_log2 & 255);
    unsigned long long *lu = reinterpret_cast<unsigned long long *>(ls);

    const int lane = tid & 63;
    const int pix = lane & 15, q = lane >> 4;
    const int wv_in_block = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t wave = (int64_t)wv_in_block * gridDim.x + blockIdx.x;   // units dealt workgroup-major: a small launch spreads over all CUs
    const int64_t nwaves = (int64_t)gridDim.x * kOneWaves;
    const int64_t nunits = (n + 31) / 32;

    half8 wreg[NB];
    f32x4 breg[FOLD ? 1 : NB];
#pragma unroll
    for (int b = 0; b < NB; b++) {
        wreg[b] = frag_l[(b * 2) * 64 + lane];
        if constexpr (!FOLD) breg[b] = bias_l[b * 64 + lane];
    }
    const half8 *wlow = frag_l;
    // the row side of the bias slots: pair 3 of every fp16 operand
    half2_t bias_mul;
    bias_mul[0] = q == 0 ? (_Float16)64.f : (q == 1 ? (_Float16)6.103515625e-05f : (_Float16)0.f);
    bias_mul[1] = q == 0 ? (_Float16)0.03125f : (q == 1 ? (_Float16)65504.f : (_Float16)0.f);

    unsigned loff[NP];
    float mus[NP][2];
#pragma unroll
    for (int p = 0; p < NP; p++) {
        int ch = q * CPL + 2 * p;
        if (ch > c - 2) ch = c - 2;
        loff[p] = (unsigned)((pix * ldx + ch) * (int64_t)sizeof(T));
        mus[p][0] = hdr->mu_s[ch];
        mus[p][1] = hdr->mu_s[ch + 1];
    }
    const int64_t tile_bytes = 16 * ldx * (int64_t)sizeof(T);

    typedef typename Pair<T>::type P2;
    typedef P2 RowSet[kOneTiles][NP];
    RowSet rows_a, rows_b;
    auto load_unit = [&](int64_t u, RowSet &raw) {
        int64_t row0 = u * 32;
        if (row0 > n - 32) row0 = n - 32;
        const char *gb = reinterpret_cast<const char *>(x) + row0 * ldx * (int64_t)sizeof(T);
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(gb), (short)0,
                                                                              (int)(32 * ldx * (int64_t)sizeof(T)), 0x00020000);
#pragma unroll
        for (int t = 0; t < kOneTiles; t++) {
            const int soff = (int)(t * tile_bytes);
#pragma unroll
            for (int p = 0; p < NP; p++) {
                if constexpr (sizeof(T) == 2) {
                    const unsigned v = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)loff[p], soff, 0);
                    const half2_t h = __builtin_bit_cast(half2_t, v);
                    raw[t][p].x = h[0];
                    raw[t][p].y = h[1];
                } else if constexpr (sizeof(T) == 4) {
                    const uint2v v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)loff[p], soff, 0);
                    raw[t][p].x = __uint_as_float(v[0]);
                    raw[t][p].y = __uint_as_float(v[1]);
                } else {
                    typedef unsigned uint4v __attribute__((ext_vector_type(4)));
                    const uint4v v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)loff[p], soff, 0);
                    raw[t][p].x = __longlong_as_double(((long long)v[1] << 32) | v[0]);
                    raw[t][p].y = __longlong_as_double(((long long)v[3] << 32) | v[2]);
                }
            }
        }
    };

    int64_t u = wave;
    if (u < nunits) load_unit(u, rows_a);
    // One trip over 32 rows.  mode 0: unit u's rows are in `raw` (the next unit's are requested into `nxt`), stage 1 only, the rows
    // it cannot vouch for go to the wave's queue.  mode 1: `raw` holds `full_rows` rows gathered from the end of that queue
    // (lane (t, pix) <-> entry s1_n - full_rows + 16 t + pix), searched with all three terms; what is still unsure then is
    // settled exactly after the loop (workgroup queue).
    auto trip = [&](RowSet &raw, RowSet &nxt, auto mode_tag, unsigned full_rows) {
        constexpr bool FULL = decltype(mode_tag)::value != 0;
        half8 bh[kOneTiles];
        float ss[kOneTiles];
        // x' = fl(x * scale - mu_s): one rounding (binary64 rows: formed in binary64, then rounded once more)
        auto centred = [&](int t, int p, float &xs0, float &xs1) {
            if constexpr (sizeof(T) == 8) {
                xs0 = (float)__builtin_fma((double)raw[t][p].x, (double)scale, -(double)mus[p][0]);
                xs1 = (float)__builtin_fma((double)raw[t][p].y, (double)scale, -(double)mus[p][1]);
            } else {
                xs0 = fmaf((float)raw[t][p].x, scale, -mus[p][0]);
                xs1 = fmaf((float)raw[t][p].y, scale, -mus[p][1]);
            }
        };
#pragma unroll
        for (int t = 0; t < kOneTiles; t++) {
            float acc2 = 0.f;
#pragma unroll
            for (int p = 0; p < (FOLD ? 3 : 4); p++) {
                half2_t h2 = {(_Float16)0, (_Float16)0};
                if (p < NP) {
                    float xs0, xs1;
                    centred(t, p < NP ? p : 0, xs0, xs1);
                    h2[0] = (_Float16)xs0;
                    h2[1] = (_Float16)xs1;
                    acc2 = __builtin_amdgcn_fdot2(h2, h2, acc2, false);
                }
                bh[t][2 * p] = h2[0];
                bh[t][2 * p + 1] = h2[1];
            }
            if constexpr (FOLD) {
                bh[t][6] = bias_mul[0];
                bh[t][7] = bias_mul[1];
            }
            ss[t] = acc2;
        }
        if constexpr (!FULL) {
            int64_t unext = u + nwaves;
            if (unext > nunits - 1) unext = nunits - 1;   // harmless re-read on the last trip
            load_unit(unext, nxt);
        }
        auto absorb = [&](float &m1, float &m2, const f32x4 &acc, int b) {
#if PXSOM_ONE_ABL & 2
            m1 = fmaxf(m1, pack_idx(acc[0], (unsigned)(b * 4), idx_mask));   // (timing build: no top-2)
            m2 = fminf(m2, acc[1] + acc[2] + acc[3]);
            return;
#endif
            if (b < NB - 1) {
                top2_quad(m1, m2, pack_idx(acc[0], (unsigned)(b * 4 + 0), idx_mask), pack_idx(acc[1], (unsigned)(b * 4 + 1), idx_mask),
                          pack_idx(acc[2], (unsigned)(b * 4 + 2), idx_mask), pack_idx(acc[3], (unsigned)(b * 4 + 3), idx_mask));
            } else {   // last block: register 0 holds its real nodes (node_of_row)
                const float p0 = pack_idx(acc[0], (unsigned)(b * 4 + 0), idx_mask);
                m2 = __builtin_amdgcn_fmed3f(m1, m2, p0);
                m1 = fmaxf(m1, p0);
            }
        };
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        float m1[kOneTiles], m2[kOneTiles];
#pragma unroll
        for (int t = 0; t < kOneTiles; t++) m1[t] = m2[t] = kNegBig;
        if constexpr (FULL) {
#pragma unroll
            for (int t = 0; t < kOneTiles; t++) {
                half8 bl;
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    half2_t l2 = {(_Float16)0, (_Float16)0};
                    if (p < NP) {
                        float xs0, xs1;
                        centred(t, p < NP ? p : 0, xs0, xs1);
                        l2[0] = (_Float16)(xs0 - (float)bh[t][2 * p]);
                        l2[1] = (_Float16)(xs1 - (float)bh[t][2 * p + 1]);
                    }
                    bl[2 * p] = l2[0];
                    bl[2 * p + 1] = l2[1];
                }
#pragma unroll
                for (int b = 0; b < NB; b++) {
                    f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[b], bh[t], FOLD ? zero4 : breg[FOLD ? 0 : b], 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[b], bl, acc, 0, 0, 0);   // (bl's bias slots are zero)
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlow[(b * 2 + 1) * 64 + lane], bh[t], acc, 0, 0, 0);   // (Wl's too)
                    absorb(m1[t], m2[t], acc, b);
                }
            }
        } else {
#pragma unroll
            for (int b = 0; b < NB; b++) {
                f32x4 acc[kOneTiles];
#pragma unroll
                for (int t = 0; t < kOneTiles; t++)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[b], bh[t], FOLD ? zero4 : breg[FOLD ? 0 : b], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < kOneTiles; t++) absorb(m1[t], m2[t], acc[t], b);
            }
        }
        // merge of the four lane groups that share a pixel: v_permlane16_swap(A = tile 0's, B = tile 1's) leaves {q0, q1} of
        // tile 0 in lane row 0, of tile 1 in row 1, {q2, q3} of tile 0 in row 2, of tile 1 in row 3; v_permlane32_swap of the
        // partial results with themselves then brings the two halves together in every row: even lane rows end with tile 0's
        // merged result, odd rows with tile 1's -- lane (q, pix) owns row 16 (q & 1) + pix of the unit, rows 2, 3 mirror 0, 1
        float a1, a2, s2;
        {
            const float t0 = __uint_as_float(__float_as_uint(m1[0]) | ((unsigned)q << 5));
            const float t1 = __uint_as_float(__float_as_uint(m1[1]) | ((unsigned)q << 5));
            uint2v r1 = __builtin_amdgcn_permlane16_swap(__float_as_uint(t0), __float_as_uint(t1), false, false);
            uint2v r2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(m2[0]), __float_as_uint(m2[1]), false, false);
            uint2v rs = __builtin_amdgcn_permlane16_swap(__float_as_uint(ss[0]), __float_as_uint(ss[1]), false, false);
            float a = __uint_as_float(r1[0]), b = __uint_as_float(r1[1]);
            const float p1 = fmaxf(a, b);
            const float p2 = fmaxf(fmaxf(fminf(a, b), __uint_as_float(r2[0])), __uint_as_float(r2[1]));
            const float ps = __uint_as_float(rs[0]) + __uint_as_float(rs[1]);
            r1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(p1), __float_as_uint(p1), false, false);
            r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(p2), __float_as_uint(p2), false, false);
            rs = __builtin_amdgcn_permlane32_swap(__float_as_uint(ps), __float_as_uint(ps), false, false);
            a = __uint_as_float(r1[0]);
            b = __uint_as_float(r1[1]);
            a1 = fmaxf(a, b);
            a2 = fmaxf(fmaxf(fminf(a, b), __uint_as_float(r2[0])), __uint_as_float(r2[1]));
            s2 = __uint_as_float(rs[0]) + __uint_as_float(rs[1]);
        }
        const float xn = __builtin_amdgcn_sqrtf(s2) * 1.001f;
        const unsigned sbits = __float_as_uint(s2);
        const unsigned unfit = (unsigned)!(xn < x_limit) | (unsigned)((sbits & 0x7f800000u) == 0x7f800000u) | (unsigned)force_exact;
        const float trel = FULL ? tol_rel : tol_rel_coarse;
        const float tol = trel * (xn * wn_max + 0.5f * wn_max * wn_max) + tol_abs * (xn + wn_max) + kTolFloor;
        bool my_amb = (((unsigned)!((a1 - a2) > tol)) | unfit) != 0u;

        const int tile = q & 1;
        int64_t row;
        bool own, counted;   // own: this lane stores / queues the row; counted: the row belongs to this trip (mirror lanes too)
        if constexpr (FULL) {
            const int slot = tile * 16 + pix;
            counted = slot < (int)full_rows;
            row = s1_q[s1_n - full_rows + (unsigned)(counted ? slot : 0)];
            own = counted && q < 2;
        } else {
            int64_t row0 = u * 32;
            if (row0 > n - 32) row0 = n - 32;
            row = row0 + tile * 16 + pix;
            counted = row >= u * 32;   // rows of a shifted last unit that the unit before it owns are left alone
            own = counted && q < 2;
        }
        const unsigned id = __float_as_uint(a1) & node_mask;
        const unsigned wq = id >> 5, wb = (id >> 2) & 7u, wr = id & 3u;
        const unsigned real = wb == (unsigned)(NB - 1) ? 16u * wb + 4u * wr + wq : 16u * wb + 4u * wq + wr;
        if (own && !my_amb) labels[row] = (int)real + 1;
#if PXSOM_ONE_ABL & 1
        if (real == 0x7fffffffu) lu[real] = (unsigned long long)raw[0][0].x;   // (timing build: no table adds)
#else
        if constexpr (TABLE) {
            // lane (q, pix) holds channels q CPL.. of rows (t, pix), t = 0, 1; tile t's labels sit in the lane rows of its parity:
            // one swap of the value with itself hands every lane {even row's, odd row's} = {tile 0's, tile 1's}.  Unsure rows,
            // rows another trip owns and clamped channel slots go to the spare table row k: no branch.
            const unsigned mine = (my_amb || !counted) ? (unsigned)k : real;
            const uint2v lr = __builtin_amdgcn_permlane16_swap(mine, mine, false, false);
            // where the last pair of the last lane group lies past the row's end (c < 4 CPL) its sixteen lanes -- one per row of
            // the tile, whose adds would go to the spare row -- carry the row's COUNT in the first of their two adds: no separate
            // count instruction (as bmu_filter_fast since round 5)
            const bool fold = c < 4 * CPL;                      // (wave-uniform)
            const bool cnt_lane = fold && q == 3;
            // Label-coherent rows (images): as bmu_filter_fast's adds (pxsom_assign_filter_fast.h, round 5) -- where most neighbours of a
            // tile agree, inclusive prefix sums of the fixed-point words along the tile's 16 rows, then one add and one subtraction per
            // run of equal labels; decided a trip ahead on the last tile's labels, the table bit-identical either way.
            // (rows out of the wave's queue are not neighbours, the idle lanes of a short batch all carry the spare label: such a trip
            // neither takes the decision nor makes the next one)
            const bool scan_now = FULL ? false : scan_trip;
            if constexpr (FIXT && PXSOM_ADD_SCAN && !FULL) {
                const unsigned nxl = (unsigned)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)lr[kOneTiles - 1], 0x101 /* row_shl:1 */, 0xf, 0xf, false);
                scan_trip = fold && __popcll(__ballot(nxl == lr[kOneTiles - 1])) >= PXSOM_ADD_SCAN_MIN;
            }
            if constexpr (FIXT && PXSOM_ADD_SCAN)
            if (__builtin_expect(scan_now, 0)) {
#pragma unroll
                for (int t = 0; t < kOneTiles; t++) {
                    const unsigned lab = lr[t];
                    const unsigned base = __umul24(lab, (unsigned)cs), spare = __umul24((unsigned)k, (unsigned)cs);
                    const unsigned nxt = (unsigned)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)lab, 0x101 /* row_shl:1 */, 0xf, 0xf, false);
                    const bool last = pix == 15;
                    const bool ends = last || nxt != lab;
                    unsigned lo[2 * NP], hi[2 * NP];
#pragma unroll
                    for (int p = 0; p < NP; p++) {
                        const unsigned long long bx = (unsigned long long)__double_as_longlong((double)raw[t][p].x + fx.magic);
                        const unsigned long long by = (unsigned long long)__double_as_longlong((double)raw[t][p].y + fx.magic);
                        const unsigned long long b0 = (p == NP - 1 && cnt_lane) ? 1ull : bx;   // (the count lane's first word counts rows)
                        lo[2 * p] = (unsigned)b0;
                        hi[2 * p] = (unsigned)(b0 >> 32);
                        lo[2 * p + 1] = (unsigned)by;
                        hi[2 * p + 1] = (unsigned)(by >> 32);
                    }
#define PXSOM_SCAN_STEP(SHR)                                                                                                         \
    _Pragma("unroll") for (int j = 0; j < 2 * NP; j++)                                                                                \
        asm volatile("v_add_co_u32_dpp %0, vcc, %0, %0 row_shr:" #SHR " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"               \
                     "v_addc_co_u32_dpp %1, vcc, %1, %1, vcc row_shr:" #SHR " row_mask:0xf bank_mask:0xf bound_ctrl:1"                \
                     : "+v"(lo[j]), "+v"(hi[j])::"vcc");
                    PXSOM_SCAN_STEP(1)
                    PXSOM_SCAN_STEP(2)
                    PXSOM_SCAN_STEP(4)
                    PXSOM_SCAN_STEP(8)
#undef PXSOM_SCAN_STEP
                    if (ends) {
                        const unsigned nlab = nxt < (unsigned)k ? nxt : (unsigned)k;
                        const unsigned base_n = __umul24(nlab, (unsigned)cs), cnt_base = (unsigned)(k + 1) * (unsigned)cs;
#pragma unroll
                        for (int p = 0; p < NP; p++) {
                            const bool real_slot = q * CPL + 2 * p <= c - 2;
                            const unsigned off = (unsigned)(real_slot ? q * CPL + 2 * p : 0);
                            const bool counts_here = p == NP - 1 && cnt_lane;
                            const unsigned ip = (real_slot ? base : spare) + off, in = (real_slot ? base_n : spare) + off;
                            const unsigned ip0 = counts_here ? cnt_base + lab : ip, in0 = counts_here ? cnt_base + nlab : in;
                            const unsigned long long p0 = ((unsigned long long)hi[2 * p] << 32) | lo[2 * p];
                            const unsigned long long p1 = ((unsigned long long)hi[2 * p + 1] << 32) | lo[2 * p + 1];
                            __hip_atomic_fetch_add(lu + ip0, p0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            __hip_atomic_fetch_add(lu + ip + 1, p1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            if (!last) {
                                __hip_atomic_fetch_add(lu + in0, 0ull - p0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                __hip_atomic_fetch_add(lu + in + 1, 0ull - p1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            }
                        }
                    }
                }
            }
            if (__builtin_expect(!scan_now, 1))
#pragma unroll
            for (int t = 0; t < kOneTiles; t++) {
                const unsigned lab = lr[t];
                const unsigned base = __umul24(lab, (unsigned)cs), spare = __umul24((unsigned)k, (unsigned)cs);   // (v_mul_lo_u32 runs at quarter rate)
#pragma unroll
                for (int p = 0; p < NP; p++) {
                    const bool real_slot = q * CPL + 2 * p <= c - 2;
#if PXSOM_ONE_ABL & 4      // (timing build: every lane adds into words of its own -- no bank or address conflicts)
                    const unsigned idx = (unsigned)lane * 2u + (lab >> 20) + (real_slot ? 0u : (base ^ spare) >> 28);
#else
                    const unsigned idx = (real_slot ? base : spare) + (unsigned)(real_slot ? q * CPL + 2 * p : 0);
#endif
#if PXSOM_ONE_ABL & 8      // (timing build: the fixed-point values are formed, the LDS never sees them)
                    const unsigned long long v0 = (unsigned long long)__double_as_longlong((double)raw[t][p].x + fx.magic),
                                             v1 = (unsigned long long)__double_as_longlong((double)raw[t][p].y + fx.magic);
                    if ((v0 ^ v1) == 0x7ff0123456789abcull + idx) lu[0] = v0;
#else
#if PXSOM_ONE_ABL & 16     // (timing build: half of the adds)
                    if (p & 1) continue;
#endif
                    const bool counts_here = p == NP - 1 && cnt_lane;
                    const unsigned idx0 = counts_here ? (unsigned)(k + 1) * (unsigned)cs + lab : idx;
                    if constexpr (FIXT) {
                        const unsigned long long v0 = (unsigned long long)__double_as_longlong((double)raw[t][p].x + fx.magic);
                        __hip_atomic_fetch_add(lu + idx0, counts_here ? 1ull : v0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_add(lu + idx + 1, (unsigned long long)__double_as_longlong((double)raw[t][p].y + fx.magic),
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    } else {
                        __hip_atomic_fetch_add(ls + idx0, counts_here ? 1.0 : (double)raw[t][p].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_add(ls + idx + 1, (double)raw[t][p].y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
#endif
                }
                if (!fold && q == 0) {
                    if constexpr (FIXT)
                        __hip_atomic_fetch_add(lu + (size_t)(k + 1) * cs + lab, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    else
                        __hip_atomic_fetch_add(ls + (size_t)(k + 1) * cs + lab, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
#endif
        my_amb = my_amb && own;
        const unsigned long long mask = __ballot(my_amb);
        if constexpr (!FULL) {
            if (mask) {
                if (my_amb) s1_q[s1_n + (unsigned)__popcll(mask & ((1ull << lane) - 1ull))] = row;
                s1_n += (unsigned)__popcll(mask);
            }
        } else if (mask) {
            unsigned base = 0;
            if (lane == 0) base = atomicAdd(amb_n, (unsigned)__popcll(mask));
            base = __shfl(base, 0);
            const unsigned pos = base + (unsigned)__popcll(mask & ((1ull << lane) - 1ull));
            if (my_amb && pos < kOneAmbQueue) amb_q[pos] = row;
            // queue full: the rows that did not fit are settled on the spot
            unsigned long long late = __ballot(my_amb && pos >= kOneAmbQueue);
            while (late) {
                const int src = __builtin_ctzll(late);
                late &= late - 1;
                const int64_t rsrc = s1_q[s1_n - full_rows + (unsigned)src];   // (lanes q < 2: lane == slot)
                exact_row_accumulate<T, FIXT, TABLE>(x, rsrc, c, ldx, wt, k, labels, ls, lane, &fx, stats, cs, k, 1);
            }
        }
    };
    // the wave's queue, 32 rows at a time from its END (no compaction), each batch gathered into rows_b and searched in full
    // (every place the queue is emptied at comes behind a trip that has read rows_b, or behind the last trip of all)
    auto drain = [&]() {
        while (s1_n > 0u) {
            const unsigned cnt = s1_n < 32u ? s1_n : 32u;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int t = 0; t < kOneTiles; t++) {
                const int slot = t * 16 + pix;
                const T *rp = x + s1_q[s1_n - cnt + (unsigned)(slot < (int)cnt ? slot : 0)] * ldx;
#pragma unroll
                for (int p = 0; p < NP; p++) {
                    int ch = q * CPL + 2 * p;
                    if (ch > c - 2) ch = c - 2;
                    rows_b[t][p] = *reinterpret_cast<const P2 *>(rp + ch);
                }
            }
            trip(rows_b, rows_b, std::integral_constant<int, 1>{}, cnt);
            __builtin_amdgcn_wave_barrier();
            s1_n -= cnt;
        }
    };
    // (the full search stays OUT of the loop that streams the units: that loop runs until the wave's queue could overflow within
    // two more trips -- on ordinary data to the end)
    while (u < nunits) {
        while (u < nunits && s1_n <= kOneS1Queue - 64u) {
            trip(rows_a, rows_b, std::integral_constant<int, 0>{}, 0u);
            u += nwaves;
            if (u >= nunits) break;
            trip(rows_b, rows_a, std::integral_constant<int, 0>{}, 0u);
            u += nwaves;
        }
        drain();
    }
    __syncthreads();   // every wave is through its units: the workgroup's queue is complete
    {
        const unsigned queued = *amb_n < kOneAmbQueue ? *amb_n : kOneAmbQueue;   // rows past the end were settled at once
        for (unsigned i = (unsigned)(tid >> 6); i < queued; i += kOneWaves)
            exact_row_accumulate<T, FIXT, TABLE>(x, amb_q[i], c, ldx, wt, k, labels, ls, lane, &fx, stats, cs, k, 1);
    }
    __syncthreads();
    if constexpr (TMODE == 2) {
        int node = tid / c, j = tid - node * c;
        const int dnode = kOneThreads / c, dj = kOneThreads % c;
        for (int e = tid; e < k * c; e += kOneThreads) {
            const double v = ls[(size_t)node * cs + j];
            if (v != 0.0) __hip_atomic_fetch_add(stats + e, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            node += dnode;
            j += dj;
            if (j >= c) {
                j -= c;
                node++;
            }
        }
        for (int e = tid; e < k; e += kOneThreads) {
            const double v = ls[(size_t)(k + 1) * cs + e];   // counts sit behind the spare row
            if (v != 0.0) __hip_atomic_fetch_add(stats + (size_t)k * c + e, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if constexpr (FIXT) {
        int node = tid / c, j = tid - node * c;   // element e <-> (node, channel), no division per element
        const int dnode = kOneThreads / c, dj = kOneThreads % c;
        for (int e = tid; e < k * c; e += kOneThreads) {
            const unsigned long long cnt = lu[(size_t)(k + 1) * cs + node];
            if (cnt) {
                const long long units = (long long)(lu[(size_t)node * cs + j] - cnt * fx.mbits);
                if (units) __hip_atomic_fetch_add(stats + e, (double)units * fx.unit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            node += dnode;
            j += dj;
            if (j >= c) {
                j -= c;
                node++;
            }
        }
        for (int e = tid; e < k; e += kOneThreads) {
            const unsigned long long cnt = lu[(size_t)(k + 1) * cs + e];
            if (cnt) __hip_atomic_fetch_add(stats + (size_t)k * c + e, (double)cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace
}  // namespace pxsom_bmu
