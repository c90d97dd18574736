// pxsom_assign_filter.hip -- stage 2 of K7: the streaming fp16-split MFMA filter (gfx950).
//
// Compiled with -ffinite-math-only (see _build.py): the scores handled here are provably finite
// whenever a row is *not* sent to the exact path, so fmaxf/fminf lower to bare v_max_f32 /
// v_max3_f32 / v_min_f32 (no canonicalising v_max v,v in front of every bit-packed operand) and
// stay visible to the instruction scheduler.  Non-finite input rows are detected with integer
// tests on the squared-norm's bit pattern, which no floating-point assumption can fold away.
//
// Work decomposition (one wave, one iteration = 64 rows = 4 tiles of 16 pixels):
//   lane = (q << 4) | pix.  MFMA v_mfma_f32_16x16x32_f16:  D[node 16][pixel 16] += A[node][k 32] B[k][pixel]
//   A = codebook fragment of node block b (lane holds node b*16+pix... see prep), B = pixel rows:
//   lane (q, pix) holds k-slots q*8..q*8+7 = channels q*cpl .. q*cpl+cpl-1 of pixel `pix`.
//   D: lane (q, pix) holds scores of pixel `pix` for nodes 16b + 4q + r, r = 0..3.
//   Score s = X.W - |W|^2/2 with the 3-term split  Xh*Wh + Xl*Wh + Xh*Wl  (three MFMAs per block).
// Two tiles are in flight at once so each accumulator's 3-MFMA chain is interleaved with an
// independent one; the register-local top-2 (2.5 VALU ops per score) of a block overlaps the next
// block's MFMAs.
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <type_traits>

#include "pxsom_assign.h"
#include "pxsom_assign_filter_fast.h"

#ifndef PXSOM_STREAM_TP
#define PXSOM_STREAM_TP 0
#endif
#ifndef PXSOM_PACKED_TMERGE
#define PXSOM_PACKED_TMERGE 1
#endif
#ifndef PXSOM_FILTER_MODE
#define PXSOM_FILTER_MODE 0
#endif
#ifndef PXSOM_PACKED_SUPER   // packed-K filter: node blocks in super-blocks of four (0: a compare + select per block; timing builds)
#define PXSOM_PACKED_SUPER 1
#endif

namespace pxsom_bmu {
namespace {

// NB_T > 0: codebook fragments + bias live in registers (NCH_T*2*NB_T*4 + NB_T*4 VGPRs);
// NB_T == 0: fragments are streamed from the workspace (L1/L2 resident), any K.
// CPL_T > 0: compile-time channels-per-lane; 0: runtime.
// PREFETCH: the next 64-row group's loads are issued before the current group's MFMA work.
// TP: tiles in flight (2 for the register-resident shapes, 1 otherwise).
//
// Loads never branch: rows past n are clamped to row n-1 (their results are discarded) and
// channel slots past c re-read the row's last valid pair (their codebook slots are zero).
// LDSW (streamed codebook only): fragments + bias are copied into LDS once per workgroup and read from there
// (ds_read_b128) instead of from L1 / L2 for every tile set.
// BD: threads per workgroup.  256 (two workgroups per CU), or -- LDSW with a codebook that leaves room for one
// workgroup only (config 5: 125 KB of fragments) -- 512: two waves per SIMD on one LDS copy (1024 threads spill: the
// kernel holds ~110-190 VGPRs).
template <typename T, int NCH_T, int CPL_T, int NB_T, bool VEC2, bool PREFETCH, int TP, bool LDSW, int BD = 256>
__global__ __launch_bounds__(BD, BD == 256 ? 2 : 1) void bmu_filter_kernel(
    const T *__restrict__ x, int64_t n, int c, int64_t ldx, const half8 *__restrict__ wfrag,
    const f32x4 *__restrict__ bias, AssignHdr *hdr, unsigned *__restrict__ amb_list,
    int32_t *__restrict__ labels, pxsom::RowView rv)
{
    constexpr int NCH = NCH_T;
    constexpr int NFR = 2 * NCH;  // stored fragments per node block
    constexpr int CPLMAX = CPL_T > 0 ? CPL_T : 8;
    const int cpl = CPL_T > 0 ? CPL_T : hdr->cpl;
    const int nb = NB_T > 0 ? NB_T : hdr->nb;
    // only the accumulator register index r (2 bits) rides in the scores' low mantissa bits; the node block of the
    // running maximum is tracked beside it (2 VALU ops per block and tile).  With the whole (b, r) index packed,
    // K = 400 replaced 7 mantissa bits and the packing term was 60 % of the tolerance: 2.5x more rows listed.
    constexpr unsigned idx_mask = 3u;
    const float scale = hdr->scale, wn_max = hdr->wn_max, tol_abs = hdr->tol_abs, x_limit = hdr->x_limit;
    float tol_rel = hdr->tol_rel;
    const bool force_exact = hdr->force_exact != 0;
    // binary16 rows: x * scale (a power of two >= 1) IS a binary16 number, so the low part of the split is exactly
    // zero: the Wh*Xl MFMA of every chunk and the conversions that feed it are dropped (XLO = false below), and with
    // them two terms of the rigorous bound -- C of the 3C products of the fp32 accumulation and the row's share of the
    // split residual (2^-19 covers Xl*Wl and the rounding of both low parts; only the codebook's, 2^-21 with the same
    // factor-of-two margin, remains): E shrinks from (2^-21 + (3C+2) 2^-24 + 2^-19 + 2^-23) to
    // (2^-21 + (2C+2) 2^-24 + 2^-21 + 2^-23) times the same norms, ~40 % at C = 40, and with it the rows listed.
    // Only a codebook whose largest entry exceeds 128 makes scale < 1, where tiny x could lose bits: the full split.
    // (a centred workspace -- binary32 / binary64 rows only, pxsom_assign.hip -- keeps the full split: x * scale - mu_s is not
    // a binary16 number)
    const bool lo_needed = sizeof(T) != 2 || scale < 1.f || hdr->centred != 0;
    // (NCH >= 2: the cross terms have an accumulator of their own, filter_accum_units_split -- the workspace's tolerance was formed
    // with the same rule, filter_accum_units_for)
    constexpr bool SPLIT = NCH >= 2;
    if (!lo_needed)
        tol_rel -= 2.5f * ((float)(SPLIT ? filter_accum_units_split(c, 3) - filter_accum_units_split(c, 2) : filter_accum_units(c, 3) - filter_accum_units(c, 2)) * 0x1p-24f +
                           (0x1p-19f - 0x1p-21f));

    const int lane = threadIdx.x & 63;
    const int pix = lane & 15, q = lane >> 4;
    constexpr int WV = BD / 64;
    const int64_t wave = (int64_t)blockIdx.x * WV + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * WV;
    const int64_t ngroups = (n + 63) / 64;

    // register-resident codebook
    half8 wreg[NB_T > 0 ? NB_T : 1][NFR];
    f32x4 breg[NB_T > 0 ? NB_T : 1];
    if constexpr (NB_T > 0) {
#pragma unroll
        for (int b = 0; b < NB_T; b++) {
#pragma unroll
            for (int s = 0; s < NFR; s++) wreg[b][s] = wfrag[(b * NFR + s) * 64 + lane];
            breg[b] = bias[b * 64 + lane];
        }
    }

    extern __shared__ __attribute__((aligned(16))) char filt_smem[];
    half8 *lfrag = reinterpret_cast<half8 *>(filt_smem);
    f32x4 *lbias = reinterpret_cast<f32x4 *>(lfrag + (size_t)nb * NFR * 64);
    if constexpr (LDSW && NB_T == 0) {
        const int nf = nb * NFR * 64;
        for (int i0 = threadIdx.x; i0 < nf; i0 += 4 * BD) {   // 4 x 16 B in flight per thread
            half8 v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = wfrag[i0 + u * BD < nf ? i0 + u * BD : 0];
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (i0 + u * BD < nf) lfrag[i0 + u * BD] = v[u];
        }
        for (int i = threadIdx.x; i < nb * 64; i += BD) lbias[i] = bias[i];
        __syncthreads();
    }

    // per-lane element offsets inside a row (clamped into the row): slot (h, i)
    int choff[NCH][CPLMAX];
#pragma unroll
    for (int h = 0; h < NCH; h++) {
#pragma unroll
        for (int i = 0; i < CPLMAX; i++) {
            int ch = h * 4 * cpl + q * cpl + (i < cpl ? i : 0);
            if constexpr (VEC2) {
                if (i & 1) ch = 0;  // unused: pairs are addressed by their even slot
                else if (ch > c - 2) ch = c - 2;
            } else {
                if (ch > c - 1) ch = c - 1;
            }
            choff[h][i] = ch;
        }
    }

    // the centring vector at this lane's channel slots, scaled (AssignHdr::mu_s: zeros when the workspace was prepared
    // without it -- the fused operation below then IS the plain product, bit for bit)
    float mus[NCH][CPLMAX];
#pragma unroll
    for (int h = 0; h < NCH; h++) {
#pragma unroll
        for (int i = 0; i < CPLMAX; i++) {
            int ch = choff[h][i];
            if constexpr (VEC2) {
                if (i & 1) ch = choff[h][i - (i & 1)] + 1;   // pairs are addressed by their even slot
            }
            mus[h][i] = hdr->mu_s[ch < kFilterMaxChannels ? ch : 0];
        }
    }

    // dst[h][i]: channel h*4*cpl + q*cpl + i of row g*64 + t*16 + pix
    auto load_tile = [&](int64_t g, int t, T(&dst)[NCH][CPLMAX]) {
        int64_t row = g * 64 + t * 16 + pix;
        if (row > n - 1) row = n - 1;
        const T *rp = x + rv.offset(row, ldx);
#pragma unroll
        for (int h = 0; h < NCH; h++) {
            if constexpr (VEC2) {
#pragma unroll
                for (int i = 0; i < CPLMAX; i += 2) {
                    const typename Pair<T>::type v =
                        *reinterpret_cast<const typename Pair<T>::type *>(rp + choff[h][i]);
                    dst[h][i] = v.x;
                    dst[h][i + 1] = v.y;
                }
            } else {
#pragma unroll
                for (int i = 0; i < CPLMAX; i++) dst[h][i] = rp[choff[h][i]];
            }
        }
    };

    // fp32 row slice -> fp16 hi / lo B-fragments + partial squared norm
    auto convert = [&](const T(&src)[NCH][CPLMAX], half8(&bh)[NCH], half8(&bl)[NCH], float &ss, auto xlo_tag) {
        constexpr bool XLO = decltype(xlo_tag)::value;
        float acc2 = 0.f;
#pragma unroll
        for (int h = 0; h < NCH; h++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                // x' = fl(x * scale - mu_s): one rounding (binary64 rows: formed in binary64, then rounded once more)
                float xf = 0.f;
                if (i < CPLMAX) {
                    if constexpr (sizeof(T) == 8)
                        xf = (float)__builtin_fma((double)src[h][i < CPLMAX ? i : 0], (double)scale, -(double)mus[h][i < CPLMAX ? i : 0]);
                    else
                        xf = fmaf((float)src[h][i < CPLMAX ? i : 0], scale, -mus[h][i < CPLMAX ? i : 0]);
                }
                acc2 = fmaf(xf, xf, acc2);
                const _Float16 hi = (_Float16)xf;
                bh[h][i] = hi;
                if constexpr (XLO) bl[h][i] = (_Float16)(xf - (float)hi);
                else bl[h][i] = (_Float16)0;
            }
        }
        ss = acc2;
    };

    T raw[PREFETCH ? kTilesPerIter : TP][NCH][CPLMAX];

    int64_t g = wave;
    if constexpr (PREFETCH) {
        if (g < ngroups) {
#pragma unroll
            for (int t = 0; t < kTilesPerIter; t++) load_tile(g, t, raw[t]);
        }
    }
    auto group = [&](auto xlo_tag) {
        constexpr bool XLO = decltype(xlo_tag)::value;
        half8 bh[PREFETCH ? kTilesPerIter : TP][NCH], bl[PREFETCH ? kTilesPerIter : TP][NCH];
        float ss[PREFETCH ? kTilesPerIter : TP];
        if constexpr (PREFETCH) {
            // convert the current group's rows, then issue the next group's loads
#pragma unroll
            for (int t = 0; t < kTilesPerIter; t++) convert(raw[t], bh[t], bl[t], ss[t], xlo_tag);
            int64_t gnext = g + nwaves;
            if (gnext > ngroups - 1) gnext = ngroups - 1;  // harmless re-read on the last trip
#pragma unroll
            for (int t = 0; t < kTilesPerIter; t++) load_tile(gnext, t, raw[t]);
        }

        int my_node = 0;
        bool my_amb = false;
        float g1[kTilesPerIter], g2[kTilesPerIter], gs[kTilesPerIter];   // per tile: top-2, |X|^2 share, winner's node
        int gn[kTilesPerIter];
#pragma unroll
        for (int t0 = 0; t0 < kTilesPerIter; t0 += TP) {
            if constexpr (!PREFETCH) {
#pragma unroll
                for (int u = 0; u < TP; u++) {
                    load_tile(g, t0 + u, raw[u]);
                    convert(raw[u], bh[u], bl[u], ss[u], xlo_tag);
                }
            }
            float m1[TP], m2[TP];
            int bsel[TP];
#pragma unroll
            for (int u = 0; u < TP; u++) {
                m1[u] = m2[u] = kNegBig;
                bsel[u] = 0;
            }
            // block b's four scores into the running top-2; remember b when the maximum moved (an equal later
            // score leaves it where it was: the earlier block)
            auto absorb = [&](int u, const f32x4 &a, int b) {
                const float before = m1[u];
                consume(m1[u], m2[u], a, 0, idx_mask);
                bsel[u] = m1[u] != before ? b : bsel[u];
            };

            if constexpr (NB_T > 0) {
#pragma unroll
                for (int b = 0; b < NB_T; b++) {
                    f32x4 acc[TP], accx[SPLIT ? TP : 1];
#pragma unroll
                    for (int u = 0; u < TP; u++) acc[u] = breg[b];
#pragma unroll
                    for (int u = 0; u < (SPLIT ? TP : 1); u++) accx[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                    // per chunk: Wh*Xh + Wh*Xl + Wl*Xh, the TP chains interleaved (SPLIT: the cross terms in accx)
#pragma unroll
                    for (int h = 0; h < NCH; h++) {
#pragma unroll
                        for (int u = 0; u < TP; u++) {
                            const int sl = PREFETCH ? t0 + u : u;
                            acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[b][2 * h], bh[sl][h], acc[u], 0, 0, 0);
                        }
#pragma unroll
                        for (int u = 0; u < TP; u++) {
                            const int sl = PREFETCH ? t0 + u : u;
                            if constexpr (XLO) {
                                if constexpr (SPLIT) accx[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[b][2 * h], bl[sl][h], accx[u], 0, 0, 0);
                                else acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[b][2 * h], bl[sl][h], acc[u], 0, 0, 0);
                            }
                        }
#pragma unroll
                        for (int u = 0; u < TP; u++) {
                            const int sl = PREFETCH ? t0 + u : u;
                            if constexpr (SPLIT) accx[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[b][2 * h + 1], bh[sl][h], accx[u], 0, 0, 0);
                            else acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[b][2 * h + 1], bh[sl][h], acc[u], 0, 0, 0);
                        }
                    }
                    if constexpr (SPLIT) {
#pragma unroll
                        for (int u = 0; u < TP; u++)
#pragma unroll
                            for (int r = 0; r < 4; r++) acc[u][r] = acc[u][r] + accx[u][r];
                    }
#pragma unroll
                    for (int u = 0; u < TP; u++) absorb(u, acc[u], b);
                }
            } else {
                for (int b = 0; b < nb; b++) {
                    f32x4 acc[TP], accx[SPLIT ? TP : 1];
                    f32x4 bv;
                    if constexpr (LDSW) bv = lbias[b * 64 + lane];
                    else bv = bias[b * 64 + lane];
#pragma unroll
                    for (int u = 0; u < TP; u++) acc[u] = bv;
#pragma unroll
                    for (int u = 0; u < (SPLIT ? TP : 1); u++) accx[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int h = 0; h < NCH; h++) {
                        half8 wh, wl;
                        if constexpr (LDSW) {
                            wh = lfrag[(b * NFR + 2 * h) * 64 + lane];
                            wl = lfrag[(b * NFR + 2 * h + 1) * 64 + lane];
                        } else {
                            wh = wfrag[(b * NFR + 2 * h) * 64 + lane];
                            wl = wfrag[(b * NFR + 2 * h + 1) * 64 + lane];
                        }
#pragma unroll
                        for (int u = 0; u < TP; u++) {
                            const int sl = PREFETCH ? t0 + u : u;
                            acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, bh[sl][h], acc[u], 0, 0, 0);
                            if constexpr (SPLIT) {
                                if constexpr (XLO) accx[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, bl[sl][h], accx[u], 0, 0, 0);
                                accx[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, bh[sl][h], accx[u], 0, 0, 0);
                            } else {
                                if constexpr (XLO) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, bl[sl][h], acc[u], 0, 0, 0);
                                acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, bh[sl][h], acc[u], 0, 0, 0);
                            }
                        }
                    }
                    if constexpr (SPLIT) {
#pragma unroll
                        for (int u = 0; u < TP; u++)
#pragma unroll
                            for (int r = 0; r < 4; r++) acc[u][r] = acc[u][r] + accx[u][r];
                    }
#pragma unroll
                    for (int u = 0; u < TP; u++) absorb(u, acc[u], b);
                }
            }
#pragma unroll
            for (int u = 0; u < TP; u++) {   // the tile's result waits for the other three: one merge for the group below
                const unsigned bb = (unsigned)bsel[u], r = __float_as_uint(m1[u]) & idx_mask;
                g1[t0 + u] = m1[u];
                g2[t0 + u] = m2[u];
                gs[t0 + u] = ss[PREFETCH ? t0 + u : u];
                gn[t0 + u] = (int)((int)bb == nb - 1 ? (bb << 4) | (r << 2) | (unsigned)q : (bb << 4) | ((unsigned)q << 2) | r);
            }
        }
        {
            // Transposing merge of the group's four tiles (the register-resident filter's, with the node index beside the
            // score -- the only perturbation of the scores is the 2-bit register index): v_permlane16_swap(A, B) exchanges
            // the odd lane rows of A with the even ones of B, v_permlane32_swap the lane halves; afterwards lane row q holds
            // tile q's merged result.  12 swaps per 64 rows where the tile-by-tile merge took 32.
            auto tmerge = [&](int x, int y, bool wide) {   // tiles x, y -> slot x
                uint2v r1, r2, rs, rn;
                if (wide) {
                    r1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(g1[x]), __float_as_uint(g1[y]), false, false);
                    r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(g2[x]), __float_as_uint(g2[y]), false, false);
                    rs = __builtin_amdgcn_permlane32_swap(__float_as_uint(gs[x]), __float_as_uint(gs[y]), false, false);
                    rn = __builtin_amdgcn_permlane32_swap((unsigned)gn[x], (unsigned)gn[y], false, false);
                } else {
                    r1 = __builtin_amdgcn_permlane16_swap(__float_as_uint(g1[x]), __float_as_uint(g1[y]), false, false);
                    r2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(g2[x]), __float_as_uint(g2[y]), false, false);
                    rs = __builtin_amdgcn_permlane16_swap(__float_as_uint(gs[x]), __float_as_uint(gs[y]), false, false);
                    rn = __builtin_amdgcn_permlane16_swap((unsigned)gn[x], (unsigned)gn[y], false, false);
                }
                const float ea = __uint_as_float(r1[0]), eb = __uint_as_float(r1[1]);
                const int na = (int)rn[0], nb_ = (int)rn[1];
                const bool take_b = eb > ea || (eb == ea && nb_ < na);
                g2[x] = fmaxf(fmaxf(fminf(ea, eb), __uint_as_float(r2[0])), __uint_as_float(r2[1]));
                g1[x] = take_b ? eb : ea;
                gn[x] = take_b ? nb_ : na;
                gs[x] = __uint_as_float(rs[0]) + __uint_as_float(rs[1]);
            };
            tmerge(0, 1, false);
            tmerge(2, 3, false);
            tmerge(0, 2, true);
            // |X| up to 2^-20 relative; integer test catches NaN/Inf rows under finite-math
            const float xn = __builtin_amdgcn_sqrtf(gs[0]) * 1.000001f;
            const float tol = tol_rel * (xn * wn_max + 0.5f * wn_max * wn_max) + tol_abs * (xn + wn_max) + kTolFloor;
            // launder the bits through an empty asm: under -ffinite-math-only the optimiser would
            // otherwise fold this exponent test (it recognises it as an is-nan-or-inf query) to false
            unsigned sbits = __float_as_uint(gs[0]);
            asm volatile("" : "+v"(sbits));
            const bool nonfinite = (sbits & 0x7f800000u) == 0x7f800000u;
            my_amb = !((g1[0] - g2[0]) > tol) || !(xn < x_limit) || nonfinite || force_exact;
            my_node = gn[0];
        }
        // lane (q, pix) now owns row g*64 + q*16 + pix == g*64 + lane
        const int64_t row = g * 64 + lane;
        const bool valid = row < n;
        if (valid) labels[row] = my_node + 1;
        const bool push = valid && my_amb;
        const unsigned long long mask = __ballot(push);
        if (mask) {
            unsigned base = 0;
            if (lane == 0) base = atomicAdd(&hdr->amb_count, (unsigned)__popcll(mask));
            base = __shfl(base, 0);
            if (push) amb_list[base + __popcll(mask & ((1ull << lane) - 1ull))] = (unsigned)row;
        }
    };
    for (; g < ngroups; g += nwaves) {
        if constexpr (sizeof(T) == 2) {
            if (lo_needed) group(std::true_type{});
            else group(std::false_type{});
        } else {
            group(std::true_type{});
        }
    }
}

template <typename T, int CPL, int NB, int RU>
void launch_fast(const T *x, int64_t n, int c, int64_t ldx, char *ws, const Layout &L, int32_t *labels,
                 hipStream_t st)
{
    // (timing builds: -DPXSOM_FILTER_MODE=1 streams the rows without MFMA / top-2, 2 re-reads one cache-hot group)
    auto kern = bmu_filter_fast<T, CPL, NB, RU, PXSOM_FILTER_MODE, false>;
    static pxsom::PerDevice<int> bpc_on;
    int &bpc = bpc_on.here();
    if (bpc == 0) {
        int nbk = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nbk, kern, 256, 0) != hipSuccess || nbk < 1) nbk = 2;
        bpc = nbk > 8 ? 8 : nbk;
    }
    const int64_t ngroups = (n + 63) / 64;
    int grid = (int)std::min<int64_t>((ngroups + 3) / 4, (int64_t)pxsom::device_cu_count() * bpc);
    if (grid < 1) grid = 1;
    PXSOM_TIMED_LAUNCH(kern, dim3(grid), dim3(256), 0, st, x, n, c, ldx,
                       reinterpret_cast<const half8 *>(ws + L.off_wfrag),
                       reinterpret_cast<const f32x4 *>(ws + L.off_bias), reinterpret_cast<AssignHdr *>(ws),
                       reinterpret_cast<unsigned *>(ws + L.off_list), labels, L.k, (double *)nullptr,
                       (const double *)nullptr, 0, 0, 0, FinishTables{});
}

template <typename T, int NCH, int CPL, int NB, bool VEC2, bool LDSW>
void launch_filter_variant(const T *x, int64_t n, int c, int64_t ldx, char *ws, const Layout &L, int32_t *labels,
                           hipStream_t st, size_t lds)
{
    constexpr bool PF = (NCH == 1);
    // streamed codebook (NB == 0): every fragment read serves TP tiles.  Measured (filter, ms, fragments from
    // L1 / L2): C = 40, K = 400, 4.2 M rows: TP 1 / 2 / 4 = 1.27 / 0.83 / 0.75; C = 100, K = 100, 1 M rows:
    // 0.244 / 0.188 / 0.257 (four channel chunks x four tiles of fragments no longer fit the register file)
    // (round 4, four chunks: one tile per read spills 16 dwords instead of 52 and is as fast on 1 M rows, faster on the small
    // steps of a training pass -- config 4's step 2.97 -> 2.89 ms)
    constexpr int TP = (NB > 0) ? 2 : (PXSOM_STREAM_TP > 0 ? PXSOM_STREAM_TP : (NCH <= 2 ? 4 : (NCH == 3 ? 2 : 1)));
    const int64_t ngroups = (n + 63) / 64;
    if constexpr (LDSW) {
        if (lds > 64 * 1024) {   // one workgroup per CU: make it a big one
            auto big = bmu_filter_kernel<T, NCH, CPL, NB, VEC2, PF, TP, LDSW, 512>;
            static pxsom::PerDevice<bool> raised_on;
            bool &raised = raised_on.here();
            if (!raised) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(big), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          150 * 1024);
                raised = true;
            }
            int grid = (int)std::min<int64_t>((ngroups + 7) / 8, (int64_t)pxsom::device_cu_count());
            if (grid < 1) grid = 1;
            PXSOM_TIMED_LAUNCH(big, dim3(grid), dim3(512), lds, st, x, n, c, ldx,
                               reinterpret_cast<const half8 *>(ws + L.off_wfrag),
                               reinterpret_cast<const f32x4 *>(ws + L.off_bias), reinterpret_cast<AssignHdr *>(ws),
                               reinterpret_cast<unsigned *>(ws + L.off_list), labels, pxsom::current_row_view());
            return;
        }
    }
    auto kern = bmu_filter_kernel<T, NCH, CPL, NB, VEC2, PF, TP, LDSW>;
    // persistent grid: exactly as many workgroups as are resident (VGPR- and LDS-limited), capped by the work
    static pxsom::PerDevice<int> by_regs_on;
    int &by_regs = by_regs_on.here();
    if (by_regs == 0) {
        if (LDSW)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        int nbk = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nbk, kern, 256, 0) != hipSuccess || nbk < 1) nbk = 2;
        by_regs = nbk > 8 ? 8 : nbk;
    }
    int blocks_per_cu = by_regs;
    if (lds > 0) blocks_per_cu = std::max(1, std::min<int>(by_regs, (int)((160 * 1024) / lds)));
    int grid = (int)std::min<int64_t>((ngroups + 3) / 4, (int64_t)pxsom::device_cu_count() * blocks_per_cu);
    if (grid < 1) grid = 1;
    PXSOM_TIMED_LAUNCH(kern, dim3(grid), dim3(256), lds, st, x, n, c, ldx,
                       reinterpret_cast<const half8 *>(ws + L.off_wfrag),
                       reinterpret_cast<const f32x4 *>(ws + L.off_bias), reinterpret_cast<AssignHdr *>(ws),
                       reinterpret_cast<unsigned *>(ws + L.off_list), labels, pxsom::current_row_view());
}

// ------------------------------------------------------------------------------------------------
// Packed-K streamed filter (round 3; binary16 rows, c % 8 == 0, wide codebooks -- config 5: 40 channels, 400 nodes).
// The two terms of a binary16 row's split (Wh*Xh + Wl*Xh) lie back to back along K (pxsom_assign.h packed_k): NPK MFMAs per
// node block instead of 2 per 32-channel chunk (3 instead of 4 at C = 40).  Lane (q, pix) of MFMA m holds slot group
// sg = 4 m + q: one 16-byte load of channels 8 (sg % (c / 8)) .. of its pixel's row -- the same channels feed both terms, so
// a row is read twice from L1, once from HBM.  Everything after the MFMAs (top-2 with the 2-bit register index, node block
// beside the score, transposing merge, tolerance, exact list) is the streamed kernel's.  A codebook so large that
// x * scale could lose bits (scale < 1: entries >= 256) sends every row to the exact path instead.
// ------------------------------------------------------------------------------------------------
template <int NPK, int TP, bool LDSW, int BD, bool SUPER = false>
__global__ __launch_bounds__(BD, BD == 256 ? 2 : 1) void bmu_filter_packed_kernel(
    const _Float16 *__restrict__ x, int64_t n, int c, int64_t ldx, const half8 *__restrict__ wfrag,
    const f32x4 *__restrict__ bias, AssignHdr *hdr, unsigned *__restrict__ amb_list, int32_t *__restrict__ labels,
    pxsom::RowView rv)
{
    const int nb = hdr->nb;
    // (Round 5, measured and not kept: the node block packed into the score's low bits beside the register index -- 7 bits -- instead
    // of the per-tile block index kept by a compare + select: the filter went 0.332 -> 0.362 ms on 4.2 M x 40 rows and the 2^-16
    // packing term listed 2.4 x the rows for the exact kernels: profiles/r05/packed_filter_experiments.txt)
    // Round 6: node blocks in SUPER-BLOCKS of four.  A score carries 4 bits -- which block of its super-block, which accumulator
    // register --, and the super-block of a tile's best score is kept by ONE compare + select per super-block and tile instead of one
    // per block (8 + 4 of a block's 51 vector instructions at four tiles).  Packing term of the tolerance: 2^-19 instead of 2^-21
    // (prep derived the bound for 2 bits: the difference is added here).  SUPER = false for the training steps: the crowded codebooks
    // of a pass's first steps list 30 % more rows under the wider tolerance and their exact path costs more than the filter gains
    // (config 5's shape, same box: filter 2.75 -> 2.59 ms per 33.5 M rows, training pass 3.25 -> 3.34 ms; profiles/r06/experiments.txt).
    constexpr unsigned idx_mask = (PXSOM_PACKED_SUPER && SUPER) ? 15u : 3u;
    const float scale = hdr->scale, wn_max = hdr->wn_max, tol_abs = hdr->tol_abs, x_limit = hdr->x_limit;
    // binary16 rows without the Xl terms: the bound of the streamed kernel (see there)
    const float tol_rel = hdr->tol_rel - 2.5f * ((float)(filter_accum_units(c, 3) - filter_accum_units(c, 2)) * 0x1p-24f + (0x1p-19f - 0x1p-21f)) +
                          ((PXSOM_PACKED_SUPER && SUPER) ? 2.5f * (0x1p-19f - 0x1p-21f) : 0.f);
    const bool force_exact = hdr->force_exact != 0 || scale < 1.f;
    const int lane = threadIdx.x & 63;
    const int pix = lane & 15, q = lane >> 4;
    constexpr int WV = BD / 64;
    const int64_t wave = (int64_t)blockIdx.x * WV + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * WV;
    const int64_t ngroups = (n + 63) / 64;

    extern __shared__ __attribute__((aligned(16))) char filt_smem[];
    half8 *lfrag = reinterpret_cast<half8 *>(filt_smem);
    f32x4 *lbias = reinterpret_cast<f32x4 *>(lfrag + (size_t)nb * NPK * 64);
    if constexpr (LDSW) {
        const int nf = nb * NPK * 64;
        for (int i0 = threadIdx.x; i0 < nf; i0 += 4 * BD) {
            half8 v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = wfrag[i0 + u * BD < nf ? i0 + u * BD : 0];
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (i0 + u * BD < nf) lfrag[i0 + u * BD] = v[u];
        }
        for (int i = threadIdx.x; i < nb * 64; i += BD) lbias[i] = bias[i];
        __syncthreads();
    }
    // this lane's slot groups: channel offset of its 16-byte load per MFMA (-1: a slot group past both terms)
    const int g8 = c / 8;
    int choff[NPK];
    bool first_term[NPK];
#pragma unroll
    for (int m = 0; m < NPK; m++) {
        const int sg = 4 * m + q, term = sg / g8;
        choff[m] = term < 2 ? 8 * (sg - term * g8) : -1;
        first_term[m] = term == 0;
    }
    const _Float16 hscale = (_Float16)scale;   // a power of two in [1, 2^15]: exact

    for (int64_t g = wave; g < ngroups; g += nwaves) {
        int my_node = 0;
        bool my_amb = false;
#pragma unroll
        for (int t0 = 0; t0 < kTilesPerIter; t0 += TP) {
            half8 bx[TP][NPK];
            float ss[TP];
#pragma unroll
            for (int u = 0; u < TP; u++) {
                int64_t row = g * 64 + (t0 + u) * 16 + pix;
                if (row > n - 1) row = n - 1;
                const _Float16 *rp = x + rv.offset(row, ldx);
                float acc2 = 0.f;
#pragma unroll
                for (int m = 0; m < NPK; m++) {
                    half8 v = {(_Float16)0, (_Float16)0, (_Float16)0, (_Float16)0, (_Float16)0, (_Float16)0, (_Float16)0, (_Float16)0};
                    if (choff[m] >= 0) v = *reinterpret_cast<const half8 *>(rp + choff[m]);
                    v = v * hscale;
                    if (first_term[m]) {   // |X|^2 from the first term's groups: every channel once
#pragma unroll
                        for (int i = 0; i < 8; i += 2) {
                            const half2_t h2 = {v[i], v[i + 1]};
                            acc2 = __builtin_amdgcn_fdot2(h2, h2, acc2, false);
                        }
                    }
                    bx[u][m] = v;
                }
                ss[u] = acc2;
            }
            float m1[TP], m2[TP];
            int bsel[TP];
#pragma unroll
            for (int u = 0; u < TP; u++) {
                m1[u] = m2[u] = kNegBig;
                bsel[u] = 0;
            }
            auto block = [&](int b, auto jtag) {   // node block b; its scores carry (j, r) -- j: its place in the super-block
                constexpr int J = decltype(jtag)::value;
                f32x4 acc[TP];
                const f32x4 bv = LDSW ? lbias[b * 64 + lane] : bias[b * 64 + lane];
#pragma unroll
                for (int u = 0; u < TP; u++) acc[u] = bv;
#pragma unroll
                for (int m = 0; m < NPK; m++) {
                    const half8 wf = LDSW ? lfrag[(b * NPK + m) * 64 + lane] : wfrag[(b * NPK + m) * 64 + lane];
#pragma unroll
                    for (int u = 0; u < TP; u++) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, bx[u][m], acc[u], 0, 0, 0);
                }
#pragma unroll
                for (int u = 0; u < TP; u++) consume(m1[u], m2[u], acc[u], J, idx_mask);
            };
            if constexpr (PXSOM_PACKED_SUPER && SUPER) {
                for (int sb = 0; sb * 4 < nb; sb++) {
                    float before[TP];
#pragma unroll
                    for (int u = 0; u < TP; u++) before[u] = m1[u];
                    const int b0 = sb * 4;
                    block(b0, std::integral_constant<int, 0>{});
                    if (b0 + 1 < nb) block(b0 + 1, std::integral_constant<int, 1>{});
                    if (b0 + 2 < nb) block(b0 + 2, std::integral_constant<int, 2>{});
                    if (b0 + 3 < nb) block(b0 + 3, std::integral_constant<int, 3>{});
#pragma unroll
                    for (int u = 0; u < TP; u++) bsel[u] = m1[u] != before[u] ? sb : bsel[u];
                }
            } else {
                for (int b = 0; b < nb; b++) {
                    float before[TP];
#pragma unroll
                    for (int u = 0; u < TP; u++) before[u] = m1[u];
                    block(b, std::integral_constant<int, 0>{});
#pragma unroll
                    for (int u = 0; u < TP; u++) bsel[u] = m1[u] != before[u] ? b : bsel[u];
                }
            }
            if constexpr (TP == kTilesPerIter && PXSOM_PACKED_TMERGE) {
                // Transposing merge of the four tiles at once (the register-resident filter's, with the node index beside
                // the score): v_permlane16_swap(A, B) exchanges the odd lane rows of A with the even ones of B, so with
                // A = tile 2i's values and B = tile 2i+1's a lane ends up with {own, partner} of the tile its row will own;
                // v_permlane32_swap does the same for lane halves.  Afterwards lane row q holds tile q's merged result:
                // 12 swaps per 64 rows where the tile-by-tile merge took 32.
                float a1[TP], a2[TP], s2[TP];
                int node[TP];
#pragma unroll
                for (int u = 0; u < TP; u++) {
                    a1[u] = m1[u];
                    a2[u] = m2[u];
                    s2[u] = ss[u];
                    const unsigned id = __float_as_uint(a1[u]) & idx_mask, r = id & 3u;
                    const unsigned bb = (PXSOM_PACKED_SUPER && SUPER) ? (unsigned)bsel[u] * 4u + (id >> 2) : (unsigned)bsel[u];
                    node[u] = (int)((int)bb == nb - 1 ? (bb << 4) | (r << 2) | (unsigned)q : (bb << 4) | ((unsigned)q << 2) | r);
                }
                auto tmerge = [&](int x, int y, bool wide) {   // tiles x, y -> slot x
                    uint2v r1, r2, rs, rn;
                    if (wide) {
                        r1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(a1[x]), __float_as_uint(a1[y]), false, false);
                        r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(a2[x]), __float_as_uint(a2[y]), false, false);
                        rs = __builtin_amdgcn_permlane32_swap(__float_as_uint(s2[x]), __float_as_uint(s2[y]), false, false);
                        rn = __builtin_amdgcn_permlane32_swap((unsigned)node[x], (unsigned)node[y], false, false);
                    } else {
                        r1 = __builtin_amdgcn_permlane16_swap(__float_as_uint(a1[x]), __float_as_uint(a1[y]), false, false);
                        r2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(a2[x]), __float_as_uint(a2[y]), false, false);
                        rs = __builtin_amdgcn_permlane16_swap(__float_as_uint(s2[x]), __float_as_uint(s2[y]), false, false);
                        rn = __builtin_amdgcn_permlane16_swap((unsigned)node[x], (unsigned)node[y], false, false);
                    }
                    const float ea = __uint_as_float(r1[0]), eb = __uint_as_float(r1[1]);
                    const int na = (int)rn[0], nb_ = (int)rn[1];
                    const bool take_b = eb > ea || (eb == ea && nb_ < na);
                    a2[x] = fmaxf(fmaxf(fminf(ea, eb), __uint_as_float(r2[0])), __uint_as_float(r2[1]));
                    a1[x] = take_b ? eb : ea;
                    node[x] = take_b ? nb_ : na;
                    s2[x] = __uint_as_float(rs[0]) + __uint_as_float(rs[1]);
                };
                tmerge(0, 1, false);
                tmerge(2, 3, false);
                tmerge(0, 2, true);
                {
                    const float xn = __builtin_amdgcn_sqrtf(s2[0]) * 1.000001f;
                    const float tol = tol_rel * (xn * wn_max + 0.5f * wn_max * wn_max) + tol_abs * (xn + wn_max) + kTolFloor;
                    unsigned sbits = __float_as_uint(s2[0]);
                    asm volatile("" : "+v"(sbits));
                    const bool nonfinite = (sbits & 0x7f800000u) == 0x7f800000u;
                    my_amb = !((a1[0] - a2[0]) > tol) || !(xn < x_limit) || nonfinite || force_exact;
                    my_node = node[0];
                }
            } else {
#pragma unroll
            for (int u = 0; u < TP; u++) {
                // node index beside the score through the merge of the 4 lane groups of a pixel (streamed kernel's finish)
                float a1 = m1[u], a2 = m2[u], s2 = ss[u];
                int node;
                {
                    const unsigned id = __float_as_uint(a1) & idx_mask, r = id & 3u;
                    const unsigned bb = (PXSOM_PACKED_SUPER && SUPER) ? (unsigned)bsel[u] * 4u + (id >> 2) : (unsigned)bsel[u];
                    node = (int)((int)bb == nb - 1 ? (bb << 4) | (r << 2) | (unsigned)q : (bb << 4) | ((unsigned)q << 2) | r);
                }
                auto merge_step = [&](bool wide) {
                    const F2 e1 = wide ? xchg32(a1) : xchg16(a1), e2 = wide ? xchg32(a2) : xchg16(a2),
                             es = wide ? xchg32(s2) : xchg16(s2),
                             en = wide ? xchg32(__int_as_float(node)) : xchg16(__int_as_float(node));
                    const int na = __float_as_int(en.a), nb_ = __float_as_int(en.b);
                    const bool take_b = e1.b > e1.a || (e1.b == e1.a && nb_ < na);
                    a2 = fmaxf(fmaxf(fminf(e1.a, e1.b), e2.a), e2.b);
                    a1 = take_b ? e1.b : e1.a;
                    node = take_b ? nb_ : na;
                    s2 = es.a + es.b;
                };
                merge_step(false);
                merge_step(true);
                if (q == t0 + u) {
                    const float xn = __builtin_amdgcn_sqrtf(s2) * 1.000001f;
                    const float tol = tol_rel * (xn * wn_max + 0.5f * wn_max * wn_max) + tol_abs * (xn + wn_max) + kTolFloor;
                    unsigned sbits = __float_as_uint(s2);
                    asm volatile("" : "+v"(sbits));
                    const bool nonfinite = (sbits & 0x7f800000u) == 0x7f800000u;
                    my_amb = !((a1 - a2) > tol) || !(xn < x_limit) || nonfinite || force_exact;
                    my_node = node;
                }
            }
            }
        }
        const int64_t row = g * 64 + lane;
        const bool valid = row < n;
        if (valid) labels[row] = my_node + 1;
        const bool push = valid && my_amb;
        const unsigned long long mask = __ballot(push);
        if (mask) {
            unsigned base = 0;
            if (lane == 0) base = atomicAdd(&hdr->amb_count, (unsigned)__popcll(mask));
            base = __shfl(base, 0);
            if (push) amb_list[base + __popcll(mask & ((1ull << lane) - 1ull))] = (unsigned)row;
        }
    }
}

// (Round 5 built a two-stage form of this kernel -- the Wh MFMAs alone first, the rows they cannot vouch for searched in full from
// per-wave queues --: bit-equal labels, 0.397 ms against 0.330 for the one-stage kernel on config 5's shape, since with 400 nodes too
// many rows' top-2 gaps lie inside the 2^-11 |X'||W'| the dropped Wl term costs; profiles/r05/packed_filter_experiments.txt.  It has no
// shape where it wins and was removed in round 6.)
template <int NPK>
void launch_packed(const _Float16 *x, int64_t n, int c, int64_t ldx, char *ws, const Layout &L, int32_t *labels, hipStream_t st,
                   bool super_blocks)
{
    const size_t lds = (size_t)L.nb * (NPK + 1) * 1024;       // fragments + bias of the whole codebook
    const int64_t ngroups = (n + 63) / 64;
    const half8 *wf = reinterpret_cast<const half8 *>(ws + L.off_wfrag);
    const f32x4 *bi = reinterpret_cast<const f32x4 *>(ws + L.off_bias);
    AssignHdr *hd = reinterpret_cast<AssignHdr *>(ws);
    unsigned *al = reinterpret_cast<unsigned *>(ws + L.off_list);
    const int cus = pxsom::device_cu_count();
    // one workgroup per CU (the codebook's fragments take 64 .. 150 KB of its LDS): 1024 threads -- four waves per SIMD on one
    // LDS copy; measured on config 5's shape (4.2 M rows x 40 binary16, 400 nodes) 0.332 ms against 0.370 with 512 threads:
    // matrix and vector instructions of a SIMD overlap between waves, not inside one.
    if constexpr (NPK <= 4)
    if (lds <= 150 * 1024 && lds > 64 * 1024) {   // (five chunks and more spill at 128 VGPRs)
        auto kern = super_blocks ? bmu_filter_packed_kernel<NPK, 4, true, 1024, true> : bmu_filter_packed_kernel<NPK, 4, true, 1024, false>;
        static pxsom::PerDevice<bool> raised_on;
        bool &raised = raised_on.here();
        if (!raised) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            raised = true;
        }
        const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((ngroups + 15) / 16, cus));
        PXSOM_TIMED_LAUNCH(kern, dim3(grid), dim3(1024), lds, st, x, n, c, ldx, wf, bi, hd, al, labels, pxsom::current_row_view());
        return;
    }
    if (lds <= 150 * 1024 && lds > 64 * 1024) {          // one workgroup per CU: a big one (two waves per SIMD on one LDS copy)
        auto kern = bmu_filter_packed_kernel<NPK, 4, true, 512>;
        static pxsom::PerDevice<bool> raised_on;
        bool &raised = raised_on.here();
        if (!raised) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            raised = true;
        }
        const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((ngroups + 7) / 8, cus));
        PXSOM_TIMED_LAUNCH(kern, dim3(grid), dim3(512), lds, st, x, n, c, ldx, wf, bi, hd, al, labels, pxsom::current_row_view());
    } else if (lds <= 64 * 1024) {
        auto kern = bmu_filter_packed_kernel<NPK, 4, true, 256>;
        const int per_cu = std::max(1, std::min(2, (int)((160 * 1024) / lds)));
        const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((ngroups + 3) / 4, (int64_t)cus * per_cu));
        PXSOM_TIMED_LAUNCH(kern, dim3(grid), dim3(256), lds, st, x, n, c, ldx, wf, bi, hd, al, labels, pxsom::current_row_view());
    } else {                                              // fragments from L1 / L2
        auto kern = bmu_filter_packed_kernel<NPK, 4, false, 256>;
        const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((ngroups + 3) / 4, (int64_t)cus * 2));
        PXSOM_TIMED_LAUNCH(kern, dim3(grid), dim3(256), 0, st, x, n, c, ldx, wf, bi, hd, al, labels, pxsom::current_row_view());
    }
}

template <typename T, int NCH, int CPL, int NB, bool VEC2>
void launch_filter(const T *x, int64_t n, int c, int64_t ldx, char *ws, const Layout &L, int32_t *labels,
                   hipStream_t st)
{
    // fragments + bias of the whole codebook in LDS when they fit beside nothing else (150 KB): 2 * NCH + 1
    // KB per node block
    const size_t lds = (size_t)L.nb * (2 * NCH + 1) * 1024;
    if (NB == 0 && lds <= 150 * 1024)
        launch_filter_variant<T, NCH, CPL, NB, VEC2, true>(x, n, c, ldx, ws, L, labels, st, lds);
    else
        launch_filter_variant<T, NCH, CPL, NB, VEC2, false>(x, n, c, ldx, ws, L, labels, st, 0);
}

}  // namespace

// the fast kernel addresses a lane's pairs with 32-bit byte offsets inside a 64-row group
template <typename T>
static bool tile_offsets_fit(int64_t ldx)
{
    return 64 * ldx * (int64_t)sizeof(T) < (int64_t)0x7fffffff;
}

// register-resident fast path: one channel chunk (C <= 32, even), K = 97..100 (ark's default 10x10 SOM;
// the last block's 4 nodes sit in one accumulator register -> RU = 1); pair loads need 2-element
// alignment of every row start and of the base pointer
template <typename T>
bool filter_fast_path(const T *x, int64_t n, int c, int64_t ldx, const Layout &L)
{
    const bool vec2 = (c % 2 == 0) && (ldx % 2 == 0) && (reinterpret_cast<uintptr_t>(x) % (2 * sizeof(T)) == 0);
    const int nv_last = L.k - 16 * (L.nb - 1), ru = (nv_last + 3) / 4;
    return vec2 && L.nch == 1 && n >= 64 && L.nb == 7 && ru == 1 && tile_offsets_fit<T>(ldx);
}

template <typename T>
void launch_filter_any(const T *x, int64_t n, int c, int64_t ldx, char *ws, const Layout &L,
                       int32_t *labels, double *stats, const double *w, hipStream_t st, FinishTables *fin)
{
    const bool vec2 = (c % 2 == 0) && (ldx % 2 == 0) && (reinterpret_cast<uintptr_t>(x) % (2 * sizeof(T)) == 0);
    const bool fast_ok = filter_fast_path<T>(x, n, c, ldx, L);
    if (fast_ok && stats) {   // batch-rule variant (its own translation unit: default FP semantics)
        launch_filter_fast_acc<T>(x, n, c, ldx, ws, L, labels, stats, w, st, fin);
        return;
    }
    // (binary64 rows of these shapes never get here: assign_typed hands them to the two-tile kernel, pxsom_assign_onepass.h --
    // bmu_filter_fast<double> spilled and is not built any more)
    if constexpr (sizeof(T) != 8) {
        if (fast_ok) {
            if (L.cpl == 6)       // C = 18..24 (BASELINE.json configs 2/3: C = 22)
                launch_fast<T, 6, 7, 1>(x, n, c, ldx, ws, L, labels, st);
            else if (L.cpl == 8)  // C = 26..32
                launch_fast<T, 8, 7, 1>(x, n, c, ldx, ws, L, labels, st);
            else if (L.cpl == 4)  // C = 10..16
                launch_fast<T, 4, 7, 1>(x, n, c, ldx, ws, L, labels, st);
            else                  // C <= 8 (config 1)
                launch_fast<T, 2, 7, 1>(x, n, c, ldx, ws, L, labels, st);
            return;
        }
    }
    if (L.nch == 1)
        vec2 ? launch_filter<T, 1, 0, 0, true>(x, n, c, ldx, ws, L, labels, st)
             : launch_filter<T, 1, 0, 0, false>(x, n, c, ldx, ws, L, labels, st);
    else if (L.nch == 2)
        vec2 ? launch_filter<T, 2, 0, 0, true>(x, n, c, ldx, ws, L, labels, st)
             : launch_filter<T, 2, 0, 0, false>(x, n, c, ldx, ws, L, labels, st);
    else if (L.nch == 3)
        vec2 ? launch_filter<T, 3, 0, 0, true>(x, n, c, ldx, ws, L, labels, st)
             : launch_filter<T, 3, 0, 0, false>(x, n, c, ldx, ws, L, labels, st);
    else
        vec2 ? launch_filter<T, 4, 0, 0, true>(x, n, c, ldx, ws, L, labels, st)
             : launch_filter<T, 4, 0, 0, false>(x, n, c, ldx, ws, L, labels, st);
}

template void launch_filter_any<float>(const float *, int64_t, int, int64_t, char *, const Layout &, int32_t *,
                                       double *, const double *, hipStream_t, FinishTables *);
template void launch_filter_any<double>(const double *, int64_t, int, int64_t, char *, const Layout &, int32_t *,
                                        double *, const double *, hipStream_t, FinishTables *);
template void launch_filter_any<_Float16>(const _Float16 *, int64_t, int, int64_t, char *, const Layout &, int32_t *,
                                          double *, const double *, hipStream_t, FinishTables *);
void launch_filter_packed(const _Float16 *x, int64_t n, int c, int64_t ldx, char *ws, const Layout &L, int32_t *labels, hipStream_t st,
                          bool super_blocks)
{
    switch (L.npk) {   // ceil(2 c / 32) for c = 40 .. 128 (c % 8 == 0)
        case 3: launch_packed<3>(x, n, c, ldx, ws, L, labels, st, super_blocks); break;
        case 4: launch_packed<4>(x, n, c, ldx, ws, L, labels, st, super_blocks); break;
        case 5: launch_packed<5>(x, n, c, ldx, ws, L, labels, st, super_blocks); break;
        case 6: launch_packed<6>(x, n, c, ldx, ws, L, labels, st, super_blocks); break;
        case 7: launch_packed<7>(x, n, c, ldx, ws, L, labels, st, super_blocks); break;
        default: launch_packed<8>(x, n, c, ldx, ws, L, labels, st, super_blocks); break;
    }
}

template bool filter_fast_path<_Float16>(const _Float16 *, int64_t, int, int64_t, const Layout &);
template bool filter_fast_path<float>(const float *, int64_t, int, int64_t, const Layout &);
template bool filter_fast_path<double>(const double *, int64_t, int, int64_t, const Layout &);

}  // namespace pxsom_bmu
