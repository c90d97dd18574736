// pxsom_assign.h -- declarations shared by the BMU-assignment translation units.
#pragma once
#include <algorithm>
#include <cmath>

#include "pxsom_common.h"

namespace pxsom_bmu {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kHdrBytes = 1024;
constexpr int kTilesPerIter = 4;  // 4 tiles x 16 pixels = one 64-row group per wave iteration
constexpr float kNegBig = -3.0e38f;
constexpr int kFilterMaxChannels = 128;   // the MFMA filter's row width (4 chunks of 32 slots); wider rows: bmu_wide_kernel

// workspace header (one per pxsom_assign workspace)
struct AssignHdr {
    unsigned amb_count;   // rows appended to the exact list by the filter kernel
    float scale;          // power-of-two scale applied to x and w before the fp16 split
    float wn_max;         // max_k |scale*w_k|_2, rounded up
    float tol_rel;        // tol = tol_rel * (|X|*wn_max + 0.5*wn_max^2) + tol_abs*(|X| + wn_max)
    float tol_abs;
    float x_limit;        // rows with |X|_2 >= x_limit (or non-finite) go to the exact path
    int nb;               // node blocks of 16
    int nch;              // 32-slot channel chunks
    int cpl;              // channels per lane per chunk (even, <= 8)
    int idx_bits;
    int node_bits;        // bits of the node index packed into the winner at the cross-lane merge
    int force_exact;      // codebook not representable by the filter: list every row
    // Centred filter (register-resident kernels; round 4: the streamed kernel on binary32 / binary64 rows too; DESIGN.md "K7 centring").  The filter ranks the nodes by
    // X'.W' - |W'|^2 / 2 with X' = x * scale - mu_s, W' = w * scale - mu_s: the same ranking as by distance whatever mu_s
    // is, but every term of the error bound is relative to the norms of the centred vectors.  mu_s = 0: off.
    float wn_raw;         // max_k |w_k|_2 (unscaled, uncentred, rounded up): the screened exact kernel's rounding bound
    int fix_exp;          // every row the filter vouches for has |x_j| < 2^(16 - fix_exp) (fixed-point tables)
    int centred;
    float mu_s[kFilterMaxChannels];   // centring vector in scaled units (a binary32 number times the power-of-two scale: exact)
    float tol_rel_coarse; // tol_rel of the register-resident filter's first stage (Wh*Xh alone)
};
static_assert(sizeof(AssignHdr) <= kHdrBytes, "workspace header");

// The accumulation terms of the filter's |score - exact| bound (DESIGN.md "K7 error bound"), in units of 2^-24, after what
// v_mfma_f32_16x16x32_f16 was measured to do (round 5, profiles/r05/mfma_rounding.txt): an instruction takes its 32 slots as four
// groups of eight in slot order; inside a group every product is cut below 2^-24 of the group's largest product, the cut products are
// summed exactly, and the sum joins the accumulator within 2 x 2^-24 of the larger of the two (measured: 1.03 - 1.4).  So a chain of
// `terms` products per channel (Wh*Xh, Wh*Xl, Wl*Xh: 3; binary16 rows: 2; a first stage on Wh*Xh alone: 1) over NCH = ceil(C / 32)
// instructions each commits
//   * 4 NCH terms group additions, each within 2 x 2^-24 of the running magnitude |X'||W'| (1 + 2^-10) + |W'|^2 / 2
//     (+ 2: the bias the chain starts from and the rounding of |X'|^2)                                       -> filter_accum_units
//   * cuts of at most 8 x 2^-24 x max |x'_j w'_j| <= 8 x 2^-24 x |X'|_2 max|w'| per group of the Wh*Xh instructions (the cross terms'
//     groups are 2^-11 of that: the factor 1 + 2^-10), with max|w'| < 256 by the choice of the scale (every prep makes the
//     largest magnitude of the scaled codebook lie in [128, 256), or lists every row)                    -> filter_cut_abs (x |X'|)
// in place of one 2^-24 of the running magnitude per product (3C + 2), which is what round-to-nearest additions in slot order
// would commit and what the unit does not do.
// (-DPXSOM_TOL_SLOTWISE=1, timing builds: the bound of rounds 1 - 4, one rounding per product)
#ifndef PXSOM_TOL_SLOTWISE
#define PXSOM_TOL_SLOTWISE 0
#endif
__host__ __device__ inline double filter_accum_units(int c, int terms)
{
    if (PXSOM_TOL_SLOTWISE) return (double)terms * c + 2.0;
    const int nch = (c + 31) / 32;
    return 2.0 * (4.0 * nch * terms) * (1.0 + 0x1p-10) + 2.0;
}
// Round 6, the streamed filter on more than one channel chunk and the wide one-launch step: the cross terms (Wh*Xl, Wl*Xh) are
// accumulated in a register set of their OWN, started from zero, and join the Wh*Xh chain in one binary32 addition at the end.
// Their 8 NCH group additions then round against a running magnitude of at most (terms - 1) 2^-11 |X'||W'| instead of the whole
// score's, and the main chain commits only its own 4 NCH group additions:
//   2 x 4 NCH (1 + 2^-10)                       the Wh*Xh chain, as above
//   2 x 4 NCH (terms - 1) x (terms - 1) 2^-11   the cross chain (below 0.1 unit at four chunks)
//   + 1                                         the joining addition (round to nearest: half a unit of the sum, rounded up)
//   + 2                                         the bias the chain starts from and the rounding of |X'|^2
// -- 35 units at C = 100 where the single chain commits 98.  The cuts inside the groups (filter_cut_abs) are what they were.
__host__ __device__ inline double filter_accum_units_split(int c, int terms)
{
    if (PXSOM_TOL_SLOTWISE) return (double)terms * c + 2.0;
    const int nch = (c + 31) / 32;
    return 2.0 * (4.0 * nch) * (1.0 + 0x1p-10) + 2.0 * (4.0 * nch) * (terms - 1) * (terms - 1) * 0x1p-11 * (1.0 + 0x1p-10) + 1.0 + 2.0;
}
// which of the two the streamed filter (bmu_filter_kernel) runs on: the split accumulation from two channel chunks on (one chunk:
// the register-resident kernels share the workspace's tolerance and keep the single chain); packed-K fragments keep the single chain
__host__ __device__ inline bool filter_split_accumulation(int c, int npk) { return c > 32 && npk == 0; }
__host__ __device__ inline double filter_accum_units_for(int c, int terms, int npk)
{
    return filter_split_accumulation(c, npk) ? filter_accum_units_split(c, terms) : filter_accum_units(c, terms);
}
__host__ __device__ inline double filter_cut_abs(int c)
{
    if (PXSOM_TOL_SLOTWISE) return 0.0;
    const int nch = (c + 31) / 32;
    // (1 + 2^-9: the cross terms' groups, 2^-11 of the Wh*Xh ones, and the roundings of the two high parts)
    return 8.0 * (4.0 * nch) * (1.0 + 0x1p-9) * 256.0 * 0x1p-24;
}
// Every term of the bound is relative to |X'| or |W'|max: on a row that sits exactly on the centring vector of a codebook whose nodes
// all equal it (an all-zero table and an all-zero codebook: tests/test_gpu_fuzz_parity.py found it, round 6) the tolerance was 0 while
// the index bits packed into scores of +0 still made them differ by up to 127 subnormal steps (binary32 subnormals are not flushed) --
// and the kernels that do not mask duplicate nodes (wide one-launch step, BMU-only fused steps) vouched for the LAST of the equal
// nodes instead of listing the row.  A floor far above 127 x 2^-149 and far below any gap the scaled scores can show settles it:
// such rows are listed.
constexpr float kTolFloor = 0x1p-100f;
// the absolute coefficient every filter multiplies by |X'| + |W'|max: binary16 subnormal floor + the cuts
__host__ __device__ inline double filter_tol_abs(int c) { return 2.5 * (0x1p-24 * sqrt((double)c) + filter_cut_abs(c)); }

// Gain of the batch rule, 1 - (1 - alpha)^den for a whole den >= 1 (a window's row count), with q = 1 - alpha rounded once on
// the host: binary exponentiation in plain binary64 products, low bit first -- no libm call, so orc_batch_gain
// (oracle/pxsom_oracle.c, the same statements) and every kernel return the same bits whatever maths library either side links.
// `sat` = batch_gain_saturation(q), formed once per step on the host: a den >= sat has gain exactly 1 (the loop would multiply
// p <= 1 by a square of q that is itself at or below 2^-60), so such windows -- the wide ones of a pass's first steps -- skip the
// loop; below sat every factor is above 2^-60 and the product above 2^-120: no intermediate ever reaches the subnormal range.
// The loop is kept to a handful of instructions per bit (32-bit counter: a step holds fewer than 2^31 rows, sat <= den else):
// it runs on one or two waves of a latency-bound step kernel, where a wave issues one instruction every ~3.6 ns.
__host__ __device__ inline double batch_gain(double den, double q, double sat)
{
#pragma clang fp contract(off)
#ifdef PXSOM_GAIN_TIMING_EXPM1   // (timing build only: what the chain of products costs a pass against one library call)
    return -expm1(den * log(q));
#endif
    if (den >= sat) return 1.0;
    if (den >= 0x1p31) {         // (only without a saturation point below 2^31: alpha below ~1e-8)
        unsigned long long m = (unsigned long long)den;
        double p = 1.0, b = q;
        while (m) {
            p = (m & 1ull) ? p * b : p;
            m >>= 1;
            b = b * b;
        }
        return 1.0 - p;
    }
    unsigned m = (unsigned)den;
    double p = 1.0, b = q;
    while (m) {
        p = (m & 1u) ? p * b : p;
        m >>= 1;
        b = b * b;
    }
    return 1.0 - p;
}

// The smallest power of two D whose square-chain value q^D (the loop's own b) is at or below 2^-60, or 2^62 when there is none
// (q >= 1): a den >= D has a set bit at or above log2 D, the loop multiplies p <= 1 by that b or a smaller one, and 1 - p is 1.
__host__ __device__ inline double batch_gain_saturation(double q)
{
#pragma clang fp contract(off)
    double b = q, d = 1.0;
    while (b > 0x1p-60 && d < 0x1p62) {
        b = b * b;
        d = d * 2.0;
    }
    return b <= 0x1p-60 ? d : 0x1p62;
}

// Fixed-point workgroup tables (pxsom_assign_sums / pxsom_assign_means) and what becomes of the statistics buffer they are
// flushed into: the LAST workgroup through the flush (a ticket behind the statistics, cleared with them) turns [k c sums | k counts]
// into the caller's tables inside the same launch -- no finishing launch behind a 0.22 ms kernel (round 6).  `done` tells the caller
// whether the kernel that ran does this (the two-tile kernel for binary64 rows does not: the finishing kernel is launched).
struct FinishTables {
    double *sums = nullptr;        // [k, c]
    long long *counts = nullptr;   // [k]
    double *means = nullptr;       // [k, c] or NULL
    int overwrite = 0;             // 0: added into sums / counts (pxsom_assign_sums); 1: overwritten, means formed (pxsom_assign_means)
    unsigned *ticket = nullptr;    // one word behind the statistics, zero at launch
    bool done = false;
};

// the pending update a fused mini-batch step applies at its head, and its housekeeping (pxsom_batch_step.hip)
struct StepArgs {
    const double *w_in;        // [k, c] codebook the pending update applies to (W_{g-1}, or W_0 when has_update == 0)
    double *w_out;             // [k, c] receives W_g (workgroup 0); may be NULL
    const double *stats_prev;  // [k*c sums | k counts] of step g-1, all-reduced
    double *stats_zero;        // buffer cleared for step g+1 (NULL: none)
    int zero_count;
    int has_update;
    double thr, q;             // schedule of the pending update (step g-1): window threshold, 1 - alpha (batch_gain)
    double sat = 0x1p62;       // batch_gain_saturation(q)
    float tol_rel, tol_abs;    // filter tolerance coefficients (depend on c only: computed on the host)
    // two-level row view of a scheduled step (fused kernel only): row f of the step is
    // x[(f / group_w) * group_stride + (f % group_w) * ldx] -- group_w consecutive rows (the phases the step
    // takes) out of every `phases` rows; group_w == 1: a plain strided view (ldx is then ignored)
    int group_w = 1;
    int64_t group_stride = 0;
    // f / group_w without a division in the kernel (launch_step fills them): (f * group_magic) >> (32 + group_shift) for f < 2^31
    unsigned group_magic = 0;
    int group_shift = 0;
    // binary64 rows enter the statistics rounded to multiples of the run's quantum q (include/pxsom.h "Reproducible
    // statistics"): qmagic = 1.5 * 2^52 * q, (v + qmagic) - qmagic is that rounding; 0: off
    double qmagic = 0.0;
    // centring vector of the one-launch step's filter: c binary32 values in HBM (AssignHdr::mu_s before scaling); NULL: zeros
    const float *mu32 = nullptr;
    // The rule's exchange INSIDE the launch (round 5; fused kernel on a peer-to-peer communicator, pxsom_xch.h): xch_wait != 0 --
    // the pending update's statistics are the sum, in rank order, of the ranks' slots of that epoch in THIS rank's block (the
    // step waits for their flags: bounded, then NaN + the block's error word); xch_signal != 0 -- the last workgroup through the
    // flush copies this rank's statistics into its slot of EVERY rank's block and raises the flag of that epoch.
    char *const *xch_peers = nullptr;
    unsigned *xch_ticket = nullptr;
    int xch_nranks = 0, xch_rank = 0;
    unsigned long long xch_max_count = 0, xch_wait = 0, xch_signal = 0;
};

// round-half-even to the quantum behind qmagic (exact while |v| < 2^51 q)
__device__ __forceinline__ double qround(double v, double qmagic) { return qmagic != 0.0 ? (v + qmagic) - qmagic : v; }

struct Layout {
    int k, nb, nch, cpl, nsteps, idx_bits, node_bits, cp32;
    int npk = 0;   // > 0: packed-K fragments (binary16 rows, c % 8 == 0): npk MFMAs per node block, see packed_k()
    size_t off_wfrag, off_bias, off_wt, off_w32, off_list, total;
    bool has_wt() const { return off_w32 > off_wt; }
};

// Packed K axis (round 3; config 5: binary16 rows, 40 channels, 400 nodes).  binary16 rows need two terms of the split,
// Wh*Xh + Wl*Xh (x * scale is a binary16 number: Xl == 0).  The chunked layout spends one 32-slot MFMA per term and chunk
// of <= 32 channels: 4 per node block at C = 40, 62 % of the k-slots used.  Packed, the two terms lie back to back along K
// -- slots [Wh(0..C) | Wl(0..C)] against [Xh | Xh] -- and a block takes ceil(2C / 32) MFMAs: 3 at C = 40.  Slots come in
// groups of 8 (one lane's half8): group sg = 4 m + q of MFMA m, lane group q; term = sg / (C / 8), channels 8 (sg % (C / 8))...
// Needs c % 8 == 0 (a lane's 16-byte load never straddles a term or the row's end).  Worth it from 33 channels on.
inline int packed_k(int c, int k, bool rows_binary16)
{
    if (!rows_binary16 || c % 8 != 0 || c <= 32 || c > 128 || k <= 128) return 0;
    const int npk = (2 * c + 31) / 32, chunked = 2 * ((c + 31) / 32);
    return npk < chunked ? npk : 0;
}

inline Layout make_layout(int64_t n, int c, int k, int npk = 0)
{
    Layout L;
    L.npk = npk;
    L.k = k;
    L.nb = (k + 15) / 16;
    L.nch = (std::min(c, kFilterMaxChannels) + 31) / 32;   // (wide rows take no filter: the fragment regions stay small)
    int per_chunk = (c + L.nch - 1) / L.nch;          // channels per chunk
    int cpl = (per_chunk + 3) / 4;                     // per lane (4 lane groups)
    cpl = (cpl + 1) & ~1;                              // even, so float2/double2 loads stay aligned
    if (cpl > 8) cpl = 8;
    L.cpl = cpl;
    L.nsteps = npk > 0 ? npk : 2 * L.nch;  // stored fragments per node block: {Wh, Wl} per chunk, or the packed run
    // low mantissa bits of the scores that the top-2 logic replaces by an index -- its only perturbation, and a
    // term of the tolerance prep derives: the register-resident kernel (one chunk, 7 node blocks) packs a 7-bit
    // (q, b, r) id, the streamed kernel only the 2-bit accumulator register index r
    L.idx_bits = (L.nch == 1 && L.nb <= 8 && npk == 0) ? 7 : 2;
    L.node_bits = L.idx_bits;
    L.off_wfrag = kHdrBytes;
    L.off_bias = L.off_wfrag + (size_t)L.nb * L.nsteps * 64 * sizeof(half8);
    // transposed binary64 codebook [c][k] for the exact kernel when the codebook is too big for its LDS
    // (config 5: 400 x 40 = 128 KB); smaller codebooks do not use the region (kept tiny)
    L.off_wt = pxsom::align_up(L.off_bias + (size_t)L.nb * 64 * sizeof(f32x4), 256);
    const size_t wt_bytes = (size_t)k * c * sizeof(double);
    // binary32 copy [k][cp32] for the screening pass of the long-list exact kernel: rows zero-padded to one of the
    // channel counts that kernel is built for
    L.cp32 = 128;
    constexpr int kScreenBlocks[] = {13, 10, 8, 6, 5, 4, 3, 2};
    for (int cb : kScreenBlocks)
        if (c <= 8 * cb) L.cp32 = 8 * cb;
    L.off_w32 = pxsom::align_up(L.off_wt + ((wt_bytes > 64 * 1024 || c > kFilterMaxChannels) ? wt_bytes : 0), 256);
    L.off_list = pxsom::align_up(L.off_w32 + (size_t)k * L.cp32 * sizeof(float), 256);
    L.total = L.off_list + (size_t)(n > 0 ? n : 1) * sizeof(unsigned);
    return L;
}


// Node <-> MFMA row mapping.  Row m (= 4q + r in the accumulator layout) of node block b holds node
// 16b + m, except in the LAST block, whose 4x4 (q, r) index grid is transposed so that its valid
// nodes fill register r = 0 of every lane group first: node = 16b + 4r + q.  With K = 100 the last
// block's 4 nodes then sit in one accumulator register and the other three are never examined.
__host__ __device__ inline int node_of_row(int b, int m, int nb)
{
    return b == nb - 1 ? 16 * b + 4 * (m & 3) + (m >> 2) : 16 * b + m;
}

// filter stage (pxsom_assign_filter.hip, compiled with -ffinite-math-only)
// stats != nullptr (only when filter_fast_path() holds): the filter also accumulates the batch rule's
// [k*c sums | k counts] for every row it does not list
template <typename T>
void launch_filter_any(const T *x, int64_t n, int c, int64_t ldx, char *ws, const Layout &L,
                       int32_t *labels, double *stats, const double *w, hipStream_t st, FinishTables *fin = nullptr);
// the accumulating variant (pxsom_assign_filter_acc.hip): also settles its listed rows itself
template <typename T>
void launch_filter_fast_acc(const T *x, int64_t n, int c, int64_t ldx, char *ws, const Layout &L, int32_t *labels,
                            double *stats, const double *w, hipStream_t st, FinishTables *fin = nullptr);
template <typename T>
bool filter_fast_path(const T *x, int64_t n, int c, int64_t ldx, const Layout &L);
// labels only on binary64 rows of the fast path's shapes, self-contained (pxsom_assign_filter_acc.hip / pxsom_assign_onepass.h)
bool onepass_labels_route(size_t elem_bytes);
void launch_onepass_labels(const double *x, int64_t n, int c, int64_t ldx, const Layout &L, int32_t *labels, const double *w,
                           hipStream_t st);
// fused mini-batch step (pxsom_batch_step.hip): which shapes it covers, and its launch
template <typename T>
bool step_fused_shape(const T *x, int64_t n, int c, int64_t ldx, int xdim, int ydim, int64_t group_stride = 0);
template <typename T>
int launch_batch_step(const T *x, int64_t n, int c, int64_t ldx, double *stats, const StepArgs &sa,
                      int tiles_per_wave, hipStream_t st);

// pending update + codebook preparation for the generic BMU search in one launch (pxsom_batch_step.hip); returns
// false when the shape is not covered (then *rc is untouched and the caller takes the launch-per-phase route)
bool launch_update_prepare(const StepArgs &sa, int xdim, int ydim, int c, char *ws, const Layout &L, hipStream_t st, int *rc);
// One launch per BMU-only step for codebooks of up to 128 nodes x 128 channels (pxsom_batch_step_wide.hip)
template <typename T>
bool step_wide_shape(int c, int k);
int64_t step_wide_max_rows();
bool step_wide_windowed(int xdim, int ydim, int c);
template <typename T>
int launch_batch_step_wide(const T *x, int64_t n, int c, int64_t ldx, int xdim, int ydim, double *stats, const StepArgs &sa, hipStream_t st);
// the streamed filter on packed-K fragments (pxsom_assign_filter.hip)
// (super_blocks: node blocks in super-blocks of four -- the labelling calls; the training steps keep a block index per block)
void launch_filter_packed(const _Float16 *x, int64_t n, int c, int64_t ldx, char *ws, const Layout &L, int32_t *labels, hipStream_t st,
                          bool super_blocks);
// rows the packed kernel can read: 16-byte aligned rows of binary16
template <typename T>
inline bool packed_rows_ok(const T *x, int64_t ldx)
{
    return sizeof(T) == 2 && ldx % 8 == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0;
}

// pxsom_assign with the batch rule's accumulation fused in (pxsom_assign.hip).  *fused = false: the shape
// is outside the fused path, nothing was done, the caller runs assign + cluster sums separately.
// fin != NULL: the workgroup tables are 64-bit fixed point (pxsom_assign_sums: sums within 2^-38 of the codebook's largest
// magnitude per value instead of the exact binary64 sums the batch rule's tests pin; 3.4x the LDS atomic rate), and the launch may
// finish the caller's tables itself (FinishTables)
int assign_accumulate(const void *x_dev, int64_t n, int c, int64_t ldx, int dtype, const double *w_dev, int k,
                      int32_t *labels_dev, double *stats_dev, void *workspace_dev, size_t workspace_bytes,
                      hipStream_t st, bool *fused, FinishTables *fin = nullptr);
int assign_prepared(const void *x_dev, int64_t n, int c, int64_t ldx, int dtype, const double *w_dev, int k,
                    int32_t *labels_dev, void *workspace_dev, size_t workspace_bytes, hipStream_t st, int npk = 0);
int prepare_only(const double *w_dev, int c, int k, void *workspace_dev, size_t workspace_bytes,
                 double *zero_stats, hipStream_t st);

}  // namespace pxsom_bmu
