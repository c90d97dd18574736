// pxsom_assign_filter_fast.h -- the register-resident BMU filter kernel (bmu_filter_fast) and the helpers it
// shares with the generic filter.  Included by two translation units:
//   pxsom_assign_filter.hip      -ffinite-math-only, the plain filter (ACC = false): the roofline kernel
//   pxsom_assign_filter_acc.hip  default FP semantics, the batch-rule variant (ACC = true), which also
//                                resolves its listed rows itself in binary64 and must see NaNs as NaNs
#pragma once
#include <cfloat>
#include <cmath>
#include <type_traits>

#include "pxsom_assign.h"
#include "pxsom_prep.h"
#include "pxsom_wave.h"

namespace pxsom_bmu {
namespace {

template <typename T>
struct Pair;
template <>
struct Pair<float> {
    typedef float2 type;
};
template <>
struct Pair<double> {
    typedef double2 type;
};
struct half_pair {
    _Float16 x, y;
};
template <>
struct Pair<_Float16> {
    typedef half_pair type;
};

__device__ __forceinline__ float pack_idx(float v, unsigned idx, unsigned mask)
{
    return __uint_as_float((__float_as_uint(v) & ~mask) | idx);
}

// {own, partner} of a value across lanes l <-> l^16 / l^32, in lane-dependent order: only ever fed
// to symmetric functions (max/min/add), so no select is needed.  VALU only, no LDS crossbar.
struct F2 {
    float a, b;
};
__device__ __forceinline__ F2 xchg16(float v)
{
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return {__uint_as_float(r[0]), __uint_as_float(r[1])};
}
__device__ __forceinline__ F2 xchg32(float v)
{
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return {__uint_as_float(r[0]), __uint_as_float(r[1])};
}

// max / min of two scores as ONE v_med3_f32 against +-3e38.  Where the default FP semantics hold (the accumulating variants) a
// plain fmaxf on a score whose low bits were just replaced by an index costs a quieting v_max_f32 x, x, x in front of it (the
// compiler cannot know the bit pattern is no signalling NaN); the median has no such clause.  Scores are finite and below 3e38 in
// magnitude on every row whose label the filter vouches for (the others are caught by their norm).
__device__ __forceinline__ float max_of(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, -kNegBig); }
__device__ __forceinline__ float min_of(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, kNegBig); }

// running top-2 (m1 >= m2) absorbs two values per update:
//   m1' = max3(m1, a, b);   m2' = max(med3(m1, a, b), m2)
__device__ __forceinline__ void top2_pair(float &m1, float &m2, float a, float b)
{
    const float tm = __builtin_amdgcn_fmed3f(m1, a, b);
    m1 = fmaxf(fmaxf(m1, a), b);
    m2 = fmaxf(tm, m2);
}

// four values per update in FIVE operations (two top2_pair take six): the runner-up is the largest of the old one and the
// two medians, and v_max3 takes all three at once
//   t1 = med3(m1, a, b); m1' = max3(m1, a, b); t2 = med3(m1', c, d); m1'' = max3(m1', c, d); m2' = max3(m2, t1, t2)
__device__ __forceinline__ void top2_quad(float &m1, float &m2, float a, float b, float c, float d)
{
    const float t1 = __builtin_amdgcn_fmed3f(m1, a, b);
    m1 = fmaxf(fmaxf(m1, a), b);
    const float t2 = __builtin_amdgcn_fmed3f(m1, c, d);
    m1 = fmaxf(fmaxf(m1, c), d);
    m2 = fmaxf(fmaxf(m2, t1), t2);
}

__device__ __forceinline__ void consume(float &m1, float &m2, const f32x4 &acc, int b, unsigned idx_mask)
{
    const float p0 = pack_idx(acc[0], (unsigned)(b * 4 + 0), idx_mask);
    const float p1 = pack_idx(acc[1], (unsigned)(b * 4 + 1), idx_mask);
    const float p2 = pack_idx(acc[2], (unsigned)(b * 4 + 2), idx_mask);
    const float p3 = pack_idx(acc[3], (unsigned)(b * 4 + 3), idx_mask);
    top2_quad(m1, m2, p0, p1, p2, p3);
}

// ------------------------------------------------------------------------------------------------
// Fast path: one 32-slot channel chunk (C <= 32, even), K <= 128, rows 2-element aligned, n >= 64.
// Codebook fragments and bias stay in registers for the whole launch.
//   * addressing: SGPR row base + per-lane 32-bit offsets (3 VGPRs); the last, partial 64-row group
//     is shifted back to rows [n-64, n) instead of being clamped (identical labels are rewritten).
//   * conversion: hi = f16(x*s), lo = f16(x*s - hi) as v_fma_mix ops; |X|^2 from v_dot2_f32_f16.
//   * last node block: only its first RU accumulator registers hold real nodes (node_of_row).
// MODE (scripts/assign_microbench.py only): 1 = stream without MFMA/top-2, 2 = cache-hot loads.
// ------------------------------------------------------------------------------------------------
// One listed row settled by a whole wave (ACC variant): lanes <-> nodes lane and lane + 64 (K <= 128 on this
// path), the row's channels broadcast with v_readlane, distances exactly as bmu_exact_kernel / the oracle
// form them.  The winner becomes the row's label and the row is added to the workgroup's table.
#pragma clang fp contract(off)
// Fixed-point table (FIX, the one-pass labels + mean-table kernel): a value v with |v| < 2^(51-s) is added as the BIT
// PATTERN of the binary64 number v + M, M = 1.5 * 2^(52-s): in that binade an ulp is 2^-s, so bits(v + M) - bits(M) is
// round(v * 2^s) in two's complement, and the table takes ds_add_u64 (3.4x the rate of ds_add_f64 on this chip,
// scripts/ubench/lds_atomic_rate.hip) instead of a floating-point atomic.  Sums wrap modulo 2^64; the flush subtracts
// count * bits(M) and scales by 2^-s.  Per value the error is at most 2^-(s+1); s is chosen per launch so that neither a
// filter-approved row (|x * scale| < 2^16) nor a workgroup's sum can overflow (make_fixpoint).
struct FixPoint {
    double magic;                // 1.5 * 2^(52 - s)
    double limit;                // 2^(51 - s): values at or above it (and non-finite ones) bypass the table
    double unit;                 // 2^-s
    unsigned long long mbits;    // bits(magic)
};
// scale = 2^scale_exp is the filter's power-of-two scale, rows_log2 = ceil(log2(rows a workgroup can meet)).  Values the
// table takes are below 2^(16 - scale_exp) (every row the filter vouches for: |x * scale|_2 < 60000), so with
// s = 46 + scale_exp - rows_log2 a workgroup's sums stay below 2^62 units; per value the error is <= 2^-(s+1), i.e.
// 2^-(39 - rows_log2) relative to the codebook's largest magnitude (|W|max * scale is in [128, 256)).
__device__ __forceinline__ FixPoint make_fixpoint(int scale_exp, int rows_log2)
{
    // (rows_log2 below 11 would put the table's limit under the 2^(16 - scale_exp) every vouched row is only known to respect --
    // small launches -- and the vouched-row path does not test it: a workgroup's budget is sized for at least 2^11 rows)
    if (rows_log2 < 11) rows_log2 = 11;
    const int s = 46 + scale_exp - rows_log2;
    FixPoint f;
    f.magic = ldexp(1.5, 52 - s);
    f.limit = ldexp(1.0, min(51 - s, 16 - scale_exp));
    f.unit = ldexp(1.0, -s);
    f.mbits = (unsigned long long)__double_as_longlong(f.magic);
    return f;
}

// (ADD = false: the label only -- the labels-only mode of the two-tile kernel, pxsom_assign_onepass.h)
template <typename T, bool FIX = false, bool ADD = true>
__device__ __forceinline__ void exact_row_accumulate(const T *__restrict__ x, int64_t row, int c, int64_t ldx,
                                                     const double *wt, int k, int32_t *__restrict__ labels,
                                                     double *ls, int lane, const FixPoint *fx = nullptr,
                                                     double *stats = nullptr, int cs = 0, int wsj = 0, int wsn = 1,
                                                     int cnt_word = -1, unsigned long long cnt_inc = 1ull)
{
    // cnt_word >= 0 (round 6, the one-pass kernel's table): a row's count lives in word cnt_word of its table row and grows by
    // cnt_inc per row; otherwise in the region behind the spare row, by one
    // codebook element (channel j, node) at wt[j * wsj + node * wsn]: the transposed LDS copy (wsj = k, wsn = 1: the default) or
    // the row-major codebook where it lies in HBM / L2 (wsj = 1, wsn = c)
    if (wsj == 0) wsj = k;
    if (cs == 0) cs = c;   // row stride of the workgroup's table (acc_stride: padded to an odd number of words)
    const double xa = (double)x[row * ldx + (lane < c ? lane : 0)];   // c <= 32 here: lane j holds channel j
    const unsigned xlo = (unsigned)__double_as_longlong(xa), xhi = (unsigned)(__double_as_longlong(xa) >> 32);
    const int n0 = lane, n1 = lane + 64;
    const int c0 = n0 < k ? n0 : k - 1, c1 = n1 < k ? n1 : k - 1;
    double d0 = 0.0, d1 = 0.0;
    auto channel = [&](int j) {   // the row's channel j, from the lane that holds it
        return __longlong_as_double(((long long)__builtin_amdgcn_readlane(xhi, j) << 32) |
                                    (unsigned)__builtin_amdgcn_readlane(xlo, j));
    };
    int j = 0;
    for (; j + 4 <= c; j += 4) {   // the 8 LDS reads of a trip are issued together; sums stay in j order
        double wa[4], wb[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            wa[u] = wt[(size_t)(j + u) * wsj + (size_t)c0 * wsn];
            wb[u] = wt[(size_t)(j + u) * wsj + (size_t)c1 * wsn];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const double xj = channel(j + u);
            const double t0 = xj - wa[u], t1 = xj - wb[u];
            d0 += t0 * t0;
            d1 += t1 * t1;
        }
    }
    for (; j < c; j++) {
        const double xj = channel(j);
        const double t0 = xj - wt[(size_t)j * wsj + (size_t)c0 * wsn], t1 = xj - wt[(size_t)j * wsj + (size_t)c1 * wsn];
        d0 += t0 * t0;
        d1 += t1 * t1;
    }
    double best = DBL_MAX;
    int bestk = 0x7fffffff;
    const double s0 = sqrt(d0), s1 = sqrt(d1);
    if (n0 < k && s0 < best) {
        best = s0;
        bestk = n0;
    }
    if (n1 < k && s1 < best) {
        best = s1;
        bestk = n1;
    }
    const double smin = pxsom::wave_min_f64(best);
    const int win = (int)pxsom::wave_min_u32(best == smin ? (unsigned)bestk : 0xffffffffu);
    if (lane == 0) labels[row] = win == 0x7fffffff ? 0 : win + 1;   // no finite distance (NaN row): label 0
    if (ADD && win != 0x7fffffff) {
        if constexpr (FIX) {
            // a listed row may hold values the table's format cannot (it was listed for its size, perhaps): such a row
            // goes straight to the global binary64 statistics
            const bool fits = fabs(lane < c ? xa : 0.0) < fx->limit;
            unsigned long long *lu = reinterpret_cast<unsigned long long *>(ls);
            if (__ballot(!fits) == 0ull) {
                if (lane < c)
                    __hip_atomic_fetch_add(lu + (size_t)win * cs + lane, (unsigned long long)__double_as_longlong(xa + fx->magic),
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (lane == 0)
                    __hip_atomic_fetch_add(lu + (cnt_word >= 0 ? (size_t)win * cs + cnt_word : (size_t)(k + 1) * cs + win), cnt_inc,
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                if (lane < c) __hip_atomic_fetch_add(stats + (size_t)win * c + lane, xa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (lane == 0) __hip_atomic_fetch_add(stats + (size_t)k * c + win, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            if (lane < c)
                __hip_atomic_fetch_add(ls + (size_t)win * cs + lane, xa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (lane == 0)
                __hip_atomic_fetch_add(ls + (size_t)(k + 1) * cs + win, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}
#pragma clang fp contract(fast)

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef unsigned uint2v __attribute__((ext_vector_type(2)));

// Row stride of the accumulating filter's workgroup table, in 8-byte words: the channel count padded to an ODD number.  The 64
// lanes of one ds_add hit words label * stride + (lane group) * CPL + 2 p (+ 1) for 16 unrelated labels: with the natural
// stride 22 the labels spread over only 8 of the 16 double-word bank pairs (gcd(22, 16) = 2), with 23 over all of them.
__host__ __device__ inline int acc_stride(int c) { return c | 1; }
// Round 6, the fixed-point table of the one-pass kernel (FIX): a table row is 4 CPL + 1 words -- slot (lane group q, pair p,
// element e) of a lane IS word q CPL + 2 p + e of its row's table row, whatever c: words 0 .. c-1 are the channels, word c takes
// the slot that lies just past the row's end and is the row COUNT (below), the words behind it take the other slots past the end
// and are never read.  One address per tile and lane (label x stride + lane base, the pairs ride in the instruction's offset
// field), no select between a label's row and the spare one per pair, no count address: 4 address instructions per 64 rows
// instead of 36 + 20.  The stride stays odd (25 words at CPL = 6): the 16 labels of a ds_add spread over all bank pairs.
__host__ __device__ inline int acc_stride_fix(int cpl) { return 4 * cpl + 1; }
__host__ __device__ inline size_t acc_table_words(int k, int c, int cpl, bool fix)
{
    return fix ? (((size_t)(k + 1) * acc_stride_fix(cpl) + 1) & ~(size_t)1) : (((size_t)(k + 1) * (acc_stride(c) + 1) + 1) & ~(size_t)1);
}
// a * inverse(a) == 1 modulo 2^64 for odd a (Newton: the correct low bits double per step, a itself has three)
__host__ __device__ inline unsigned long long inverse_mod_2_64(unsigned long long a)
{
    unsigned long long x = a;
    for (int i = 0; i < 5; i++) x *= 2ull - a * x;
    return x;
}

// SGB_VALU > 0 forces a 1-MFMA : SGB_VALU-VALU cadence with sched_group_barrier.  Measured (bench.py,
// filter kernel): 0 -> 0.237 ms, 3 -> 0.252, 5 -> 0.249, 8 -> 0.247: the compiler's own order wins.
#ifndef SGB_VALU
#define SGB_VALU 0
#endif
#ifndef PXSOM_FAST_WGS
#define PXSOM_FAST_WGS 2
#endif
#ifndef PXSOM_PLAIN_WGS
#define PXSOM_PLAIN_WGS 2
#endif
#ifndef PXSOM_FINE_BLOCKS
#define PXSOM_FINE_BLOCKS 2
#endif
// ACC (batch-rule accumulation fused in, pxsom_batch_accumulate): every row the filter is sure of adds
// itself to a per-workgroup binary64 table [K*c sums | K counts] in LDS (ds_add_f64), flushed once with
// global atomics into `stats`.  Rows it is NOT sure of are settled on the spot by the wave that met them,
// exactly as the exact kernel would (binary64, j ascending, no contraction, sqrt, first strict minimum)
// against a transposed binary64 copy of the codebook in LDS, and added to the table too -- so a
// mini-batch step needs no exact-kernel launch.  One pass over x, one launch.
// FIX (with ACC; pxsom_assign_sums): the workgroup's table is 64-bit fixed point (FixPoint above), fix_rows_log2 =
// ceil(log2(rows a workgroup can meet)).
// Round 6: the one-pass kernel (ACC + FIX) runs as ONE workgroup of 512 threads per CU instead of two of 256 -- the same two waves
// per SIMD on one table, one codebook copy and one set of fragments: half as many workgroups prepare the codebook and flush 2 300
// binary64 atomics each into the same 144 cache lines of statistics at the launch's end (the fixed ~25 us of a 0.22 ms launch):
// 0.2311 - 0.2323 ms against 0.2406 - 0.2416 on one box (profiles/r06/experiments.txt).  -DPXSOM_ONE_WG=0: two workgroups of 256.
// (Also measured, not kept: the wave's first group requested before the prologue instead of behind it -- 0.2245 - 0.2283 against
// 0.2221 - 0.2257 ms.)
#ifndef PXSOM_ONE_WG
#define PXSOM_ONE_WG 1
#endif
#ifndef PXSOM_TIMING_ABL   // timing builds ONLY (wrong results): 1 = no flush atomics, 2 = listed rows not settled, 4 = no full search of queued rows
#define PXSOM_TIMING_ABL 0
#endif
__host__ __device__ constexpr int fast_threads(bool acc, bool fix) { return (acc && fix && PXSOM_ONE_WG) ? 512 : 256; }
// Round 6: with one workgroup per CU there is LDS to spare (70 of 160 KB): the fixed-point table exists in FOUR copies, lane (q, pix)
// adds into copy pix % 4 -- two rows of a tile that share their label meet in the same words only if they also share pix % 4, and the
// 16 lanes of an add are served together instead of one after the other (the synthetic order: 16 of 100 labels collide in 72 % of
// the instructions).  The copy is part of the lane's constant base address: no instruction in the trip; the flush adds the copies up.
#ifndef PXSOM_TABLE_COPIES
#define PXSOM_TABLE_COPIES 4
#endif
// (eight channels per lane -- C = 26 .. 32 --: two copies, four would not fit 160 KB beside the codebook copy and the fragments)
__host__ __device__ constexpr int fast_table_copies(bool acc, bool fix, int cpl)
{
    return (acc && fix && PXSOM_ONE_WG) ? (cpl <= 6 ? PXSOM_TABLE_COPIES : (PXSOM_TABLE_COPIES < 2 ? PXSOM_TABLE_COPIES : 2)) : 1;
}
#ifndef PXSOM_ADD_SCAN       // one-pass kernel: tiles whose neighbouring rows mostly share their label are summed along the row axis first
#define PXSOM_ADD_SCAN 1
#endif
#ifndef PXSOM_ADD_SCAN_MIN   // ... from this many agreeing neighbour pairs of a tile's 60 on (47: runs of five rows and longer; runs of four cost the same either way)
#define PXSOM_ADD_SCAN_MIN 47
#endif
// FOLD (FIX only): some slots of the last lane groups lie past the row's end (c < 4 CPL) -- the one at channel c then carries
// the row count; without such a slot (c == 4 CPL) the count takes an instruction of its own.
template <typename T, int CPL, int NB, int RU, int MODE, bool ACC, bool FIX = false, bool FOLD = true>
__global__ __launch_bounds__(fast_threads(ACC, FIX), (ACC && FIX && PXSOM_ONE_WG) ? 1 : (ACC ? PXSOM_FAST_WGS : PXSOM_PLAIN_WGS)) void bmu_filter_fast(
    const T *__restrict__ x, int64_t n, int c, int64_t ldx, const half8 *__restrict__ wfrag,
    const f32x4 *__restrict__ bias, AssignHdr *hdr, unsigned *__restrict__ amb_list,
    int32_t *__restrict__ labels, int k, double *__restrict__ stats, const double *__restrict__ wcodes,
    int idx_bits, int node_bits, int fix_rows_log2 = 0, FinishTables fin = FinishTables{})
{
    extern __shared__ __attribute__((aligned(16))) char acc_smem[];
    // ACC: table [(k+1)*c sums | (k+1) counts]; row k is a spare one that takes the adds of rows / channel slots that must
    // not count (listed rows, rows a previous group owns, clamped channel slots): the accumulation has no branch
    double *ls = reinterpret_cast<double *>(acc_smem);
    const int cs = FIX ? acc_stride_fix(CPL) : acc_stride(c);   // table row stride (words)
    constexpr int kCopies = fast_table_copies(ACC, FIX, CPL);
    const size_t table_words = acc_table_words(k, c, CPL, FIX);   // (of one copy)
    double *wt = ls + kCopies * table_words;   // [c][k] transposed codebook (ACC only); 16-byte aligned
    // three workgroups per CU (PXSOM_FAST_WGS = 3): no room for that copy -- the listed rows read the codebook where it lies
    constexpr bool kWtInLds = PXSOM_FAST_WGS < 3;
    const double *wx = kWtInLds ? wt : wcodes;
    const int wsj = kWtInLds ? k : 1, wsn = kWtInLds ? 1 : c;
    // ACC: rows the filter is not sure of wait in a per-workgroup queue and are settled after the group loop by
    // whichever wave is free (a mini-batch lists 0..6 rows per wave early in training: the slowest wave set the pace)
    constexpr int kThreads = fast_threads(ACC, FIX), kWaves = kThreads / 64;
    constexpr unsigned kAmbQueue = 256;
    int64_t *amb_q = nullptr;
    unsigned *amb_n = nullptr;
    // Round 4: rows the FIRST stage cannot vouch for wait in a queue of their wave and are searched in full 64 at a time
    // (full tiles, behind the group loop's body) instead of inside the trip that met them, where a tile was searched again for
    // one or two rows of its sixteen: the second stage inside the trip was 12 % of the one-pass kernel's time in execution and
    // 10 % in what its code did to the trip's schedule (profiles/r04/acc_stage2_ablation.txt)
    constexpr unsigned kS1Queue = 256;
    int64_t *s1_q = nullptr;
    unsigned s1_n = 0u;   // wave-uniform
    bool scan_trip = false;   // wave-uniform: this trip's rows are summed along the row axis before they touch the table (ACC + FIX, see there)
    if constexpr (!ACC) {   // (the plain filter has no dynamic LDS: its waves' queues are a static array)
        __shared__ long long s1_plain[kWaves * kS1Queue];
        s1_q = reinterpret_cast<int64_t *>(s1_plain) + (size_t)(threadIdx.x >> 6) * kS1Queue;
    }
    if constexpr (ACC) {
        // Every workgroup prepares the codebook for itself (no prep launch in front of a mini-batch step):
        // row-major copy in LDS -> prep_body -> fragments / bias / constants in LDS, read below exactly as
        // the plain filter reads them from the workspace.
        // the row-major copy prep_body reads lives in the table's storage (needed before the first row is added only)
        double *wrow = ls;                                                           // [k][c]
        half8 *frag_l = reinterpret_cast<half8 *>(wt + (kWtInLds ? (size_t)k * c : 0));   // [NB][2][64]
        f32x4 *bias_l = reinterpret_cast<f32x4 *>(frag_l + NB * 2 * 64);             // [NB][64]
        AssignHdr *hdr_l = reinterpret_cast<AssignHdr *>(bias_l + NB * 64);
        amb_q = reinterpret_cast<int64_t *>(reinterpret_cast<char *>(hdr_l) + kHdrBytes);   // [kAmbQueue]
        amb_n = reinterpret_cast<unsigned *>(amb_q + kAmbQueue);
        s1_q = reinterpret_cast<int64_t *>(amb_n + 4) + (size_t)(threadIdx.x >> 6) * kS1Queue;   // this wave's
        if (threadIdx.x == 0) *amb_n = 0u;
        // element e = tid + kThreads u  <->  (node, channel), advanced without a division per element
        int node = (int)threadIdx.x / c, j = (int)threadIdx.x - node * c;
        const int dnode = kThreads / c, dj = kThreads % c;
        for (int e0 = threadIdx.x; e0 < k * c; e0 += 8 * kThreads) {   // 8 L2 loads in flight per thread
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = wcodes[e0 + u * kThreads < k * c ? e0 + u * kThreads : 0];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int e = e0 + u * kThreads;
                if (e < k * c) {
                    wrow[e] = v[u];
                    if constexpr (kWtInLds) wt[j * k + node] = v[u];
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
        prep_body<kThreads, 128>(wrow, k, c, hdr_l, frag_l, bias_l, NB, 1, CPL, idx_bits, node_bits, nullptr, nullptr, 0, 0, true);
        __syncthreads();
        for (int e = threadIdx.x; e < (int)(kCopies * table_words); e += kThreads) ls[e] = 0.0;   // (the first add comes after the loads' wait)
        __syncthreads();
        wfrag = frag_l;
        bias = bias_l;
        hdr = hdr_l;
    }
    constexpr int NP = CPL / 2;  // pair loads per lane per tile
    // scores carry (b*4 + r) in their low 7 mantissa bits (inline constants <= 27: one v_and_or_b32 each);
    // OR-ing (q << 5) in yields a 7-bit id (q, b, r) that is mapped to the node once per group
    constexpr unsigned idx_mask = 127u;
    constexpr unsigned node_mask = 127u;
    static_assert(NB <= 8, "7-bit packed node index");
    const float scale = hdr->scale, wn_max = hdr->wn_max, tol_rel = hdr->tol_rel,
                tol_abs = hdr->tol_abs, x_limit = hdr->x_limit, tol_rel_coarse = hdr->tol_rel_coarse;
    const bool force_exact = hdr->force_exact != 0;
    const unsigned long long force_m = force_exact ? ~0ull : 0ull;
    // (the coefficients a shade up, the range bound a shade down: the fused form never vouches for a row the term-by-term one listed)
    const float tol_a = (tol_rel * wn_max + tol_abs) * 1.0011f, tol_b = (0.5f * tol_rel * wn_max * wn_max + tol_abs * wn_max) * 1.0001f + kTolFloor;
    const float tol_a_coarse = (tol_rel_coarse * wn_max + tol_abs) * 1.0011f,
                tol_b_coarse = (0.5f * tol_rel_coarse * wn_max * wn_max + tol_abs * wn_max) * 1.0001f + kTolFloor;
    const unsigned s2_limit_bits = __float_as_uint(fminf((x_limit / 1.001f) * (x_limit / 1.001f) * 0.9999f, 3.0e38f));
    FixPoint fx = {};
    if constexpr (FIX) {
        fx = make_fixpoint(hdr->fix_exp, fix_rows_log2 & 255);
        // Round 6: the magic number is one unit larger, so its bit pattern B is ODD.  Every slot of every vouched row then adds
        // bits(v + M) = B + round(v 2^s) to its word, the slots past the row's end (their loads return 0) B itself: the word at
        // channel c holds n B modulo 2^64 after n rows -- which is both what the flush subtracts from every channel word of the
        // row and, times the inverse of B modulo 2^64, the count n.  No lane adds anything but what it loaded.
        fx.magic += fx.unit;
        fx.mbits += 1ull;
    }
    unsigned long long *lu = reinterpret_cast<unsigned long long *>(ls);

    const int lane = threadIdx.x & 63;
    const int pix = lane & 15, q = lane >> 4;
    // ACC: groups are dealt workgroup-major (wave 0 of every workgroup first), so a mini-batch of a few hundred
    // groups spreads over all CUs -- and so do the rows its workgroups have to settle exactly
    const int wv_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // (group numbers are 32-bit -- the launchers refuse 2^37 rows and more --: their clamps and the loop tests stay on the scalar
    // unit, which has no 64-bit ordered compare)
    const int wave = ACC ? wv_in_block * (int)gridDim.x + (int)blockIdx.x : (int)blockIdx.x * kWaves + wv_in_block;
    const int nwaves = (int)gridDim.x * kWaves;
    const int ngroups = (int)((n + 63) / 64);
    const int last_shift = (int)((int64_t)ngroups * 64 - n);   // the last group is shifted back by this many rows (0: it is full)

    // the low fragments serve stage 2 only (a tile in nine on a trained codebook): the accumulating variant, which is
    // short of registers, leaves them in its LDS copy
    constexpr bool kLowInRegs = !ACC;
    half8 wreg[NB][kLowInRegs ? 2 : 1];
    const half8 *wlow = wfrag;   // (ACC: wfrag points at the workgroup's LDS copy by now)
    f32x4 breg[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) {
        wreg[b][0] = wfrag[(b * 2 + 0) * 64 + lane];
        if constexpr (kLowInRegs) wreg[b][1] = wfrag[(b * 2 + 1) * 64 + lane];
        breg[b] = bias[b * 64 + lane];
    }

    // byte offset of this lane's pair p inside a 16-row tile (channel slots past c re-read the
    // row's last valid pair: their codebook slots are zero)
    // (ACC + FIX: such a pair's offset lies past the end of every group's buffer instead -- the load returns zeros, which is what
    // the pair's table slots must add; its MFMA slots meet zero codebook entries either way)
    unsigned loff[NP];
#pragma unroll
    for (int p = 0; p < NP; p++) {
        int ch = q * CPL + 2 * p;
        const bool past = ch > c - 2;
        if (past) ch = c - 2;
        loff[p] = (unsigned)((pix * ldx + ch) * (int64_t)sizeof(T));
        if (ACC && FIX && past) loff[p] = 0x80000000u;
    }
    const int64_t tile_bytes = 16 * ldx * (int64_t)sizeof(T);
    // (Round 6, measured and not kept: a lane's first two pairs as ONE 16-byte load where both lie inside the row -- 8 requests per
    // group instead of 12 -- kernel 0.256 ms against 0.235 on the same box, interleaved; profiles/r06/experiments.txt)
    // centring vector of this lane's channels, scaled (zeros when the workspace was prepared without it)
    float mus[NP][2];
#pragma unroll
    for (int p = 0; p < NP; p++) {
        int ch = q * CPL + 2 * p;
        if (ch > c - 2) ch = c - 2;
        mus[p][0] = hdr->mu_s[ch];
        mus[p][1] = hdr->mu_s[ch + 1];
    }

    typedef typename Pair<T>::type P2;
    typedef P2 RowSet[kTilesPerIter][NP];
    // ACC: a group's rows outlive the prefetch of the next (they are added to the table after the search), so two register
    // sets take turns -- the group loop runs two trips per turn -- instead of one set being copied aside (24 v_mov per group)
    RowSet rows_a, rows_b;
    // Buffer loads: the 64-row group is a descriptor of its own (base = x + row0*ldx*sizeof(T), built
    // from wave-uniform values on the scalar unit), the tile offset rides in soffset and the lane offset in
    // voffset -- no per-load VALU address arithmetic.
    auto load_group = [&](int g, RowSet &raw) {
        if constexpr (MODE >= 2) g = wave;
        const int64_t row0 = (int64_t)g * 64 - (g == ngroups - 1 ? last_shift : 0);
        const char *gb = reinterpret_cast<const char *>(x) + row0 * ldx * (int64_t)sizeof(T);
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char *>(gb), (short)0, (int)(64 * ldx * (int64_t)sizeof(T)), 0x00020000);
#pragma unroll
        for (int t = 0; t < kTilesPerIter; t++) {
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

    int g = wave;
    if (g < ngroups) load_group(g, rows_a);
    // one trip: group g's rows are in `raw`; the next group's go to `nxt` (the same set for the plain filter, whose rows are
    // dead once converted)
    // QUEUED: `raw` holds 64 rows gathered from the wave's queue (full_rows of them are real, lane <-> queue slot);
    // FULL: every tile takes the three-term search; otherwise raw is group g and, with ACC, only stage 1 runs here.
    // mode 0: a group trip (stage 1 only, the rows it cannot vouch for are deferred); 1: the wave's queue, 64 rows in full
    auto trip = [&](RowSet &raw, RowSet &nxt, auto mode_tag, unsigned full_rows) {
        constexpr int kMode = decltype(mode_tag)::value;
        constexpr bool FULL = kMode != 0, QUEUED = kMode == 1;
        constexpr bool DEFER = !FULL && MODE != 1;   // (MODE 1: the microbenchmark's stream-only trip)
        RowSet &keep = raw;
        // (Round 6, measured neutral and not kept: the next group's loads at the very head of the trip behind a scheduling barrier --
        // the compiler sinks them behind the first three MFMAs --: 0.2341 / 0.2348 / 0.2345 / 0.2327 ms against 0.2352 / 0.2348 /
        // 0.2347 / 0.2339, interleaved on one box.  The trip does not wait for its rows.)
        half8 bh[kTilesPerIter];
        float ss[kTilesPerIter];
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
        for (int t = 0; t < kTilesPerIter; t++) {
            float acc2 = 0.f;
#pragma unroll
            for (int p = 0; p < 4; p++) {
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
            ss[t] = acc2;
        }
        // the low halves of a tile's split, on demand (the rows stay in `raw` for the whole trip: the next group's go to `nxt`)
        auto low_halves = [&](int t) {
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
            return bl;
        };
        if constexpr (!QUEUED) {
            int gnext = g + nwaves;
            if (gnext > ngroups - 1) gnext = ngroups - 1;  // harmless re-read on the last trip
            load_group(gnext, nxt);
        }

        float my_m1 = 0.f;
        // rows this trip cannot vouch for, as a lane mask: the tests below are vector compares whose results ARE such masks, and the
        // logic between them, the test "any?" and the rows' places in the queue then run on the scalar unit
        unsigned long long amb_m = 0ull;
        if constexpr (MODE == 1) {
            float acc = 0.f;
#pragma unroll
            for (int t = 0; t < kTilesPerIter; t++) acc += ss[t];
            my_m1 = __uint_as_float(__float_as_uint(acc) & node_mask);
        } else {
            // block b's scores of one tile into that tile's running top-2 (last block: only registers 0..RU-1 hold real nodes)
            auto absorb = [&](float &m1, float &m2, const f32x4 &acc, int b) {
                if (b < NB - 1 || RU == 4) {
                    top2_quad(m1, m2, pack_idx(acc[0], (unsigned)(b * 4 + 0), idx_mask), pack_idx(acc[1], (unsigned)(b * 4 + 1), idx_mask),
                              pack_idx(acc[2], (unsigned)(b * 4 + 2), idx_mask), pack_idx(acc[3], (unsigned)(b * 4 + 3), idx_mask));
                } else {
                    const float p0 = pack_idx(acc[0], (unsigned)(b * 4 + 0), idx_mask);
                    if (RU == 1) {
                        m2 = __builtin_amdgcn_fmed3f(m1, m2, p0);
                        m1 = max_of(m1, p0);
                    } else {
                        const float p1 = pack_idx(acc[1], (unsigned)(b * 4 + 1), idx_mask);
                        top2_pair(m1, m2, p0, p1);
                        if (RU == 3) {
                            const float p2 = pack_idx(acc[2], (unsigned)(b * 4 + 2), idx_mask);
                            m2 = __builtin_amdgcn_fmed3f(m1, m2, p2);
                            m1 = fmaxf(m1, p2);
                        }
                    }
                }
            };
            float tm1[kTilesPerIter], tm2[kTilesPerIter];
            // Stage 1: Wh*Xh alone, all four tiles' chains side by side -- a third of the MFMAs.  Its bound is the full one
            // with the two dropped cross terms charged to it (2^-10 |X'||W'|: AssignHdr::tol_rel_coarse); a row whose top-2
            // gap clears THAT tolerance is as sure as any.  The others wait in the wave's queue for the full search.
            float m1[kTilesPerIter], m2[kTilesPerIter];
#pragma unroll
            for (int t = 0; t < kTilesPerIter; t++) m1[t] = m2[t] = kNegBig;
            if constexpr (FULL) {
                // every tile in full: the three-term search, two node blocks' chains side by side
#pragma unroll
                for (int t = 0; t < kTilesPerIter; t++) {
                    const half8 bl = low_halves(t);
#pragma unroll
                    for (int b = 0; b < NB; b += 2) {
                        f32x4 acc[2];
#pragma unroll
                        for (int u = 0; u < 2; u++)
                            if (b + u < NB) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[b + u < NB ? b + u : b][0], bh[t], breg[b + u < NB ? b + u : b], 0, 0, 0);
#pragma unroll
                        for (int u = 0; u < 2; u++)
                            if (b + u < NB) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[b + u < NB ? b + u : b][0], bl, acc[u], 0, 0, 0);
#pragma unroll
                        for (int u = 0; u < 2; u++)
                            if (b + u < NB) {
                                half8 wl;
                                if constexpr (kLowInRegs) wl = wreg[b + u < NB ? b + u : b][1];
                                else wl = wlow[((b + u) * 2 + 1) * 64 + lane];
                                acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, bh[t], acc[u], 0, 0, 0);
                            }
#pragma unroll
                        for (int u = 0; u < 2; u++)
                            if (b + u < NB) absorb(m1[t], m2[t], acc[u], b + u);
                    }
                }
            } else {
#pragma unroll
                for (int b = 0; b < NB; b++) {
                    f32x4 acc[kTilesPerIter];
#pragma unroll
                    for (int t = 0; t < kTilesPerIter; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[b][0], bh[t], breg[b], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < kTilesPerIter; t++) absorb(m1[t], m2[t], acc[t], b);
                }
            }
#pragma unroll
            for (int t = 0; t < kTilesPerIter; t++) {
                tm1[t] = __uint_as_float(__float_as_uint(m1[t]) | ((unsigned)q << 5));   // (b*4 + r) -> id (q << 5 | b*4 + r)
                tm2[t] = m2[t];
            }
            // Transposing merge of the 4 lane groups (rows of 16 lanes) that share a pixel.
            //   v_permlane16_swap(A, B): odd rows of A <-> even rows of B.  With A = tile 2i's value and
            //   B = tile 2i+1's, even rows end up holding {own, partner} of tile 2i and odd rows those of
            //   tile 2i+1 -- in some order, which the symmetric max/min/add below do not care about.
            //   v_permlane32_swap(A, B): lanes 32-63 of A <-> lanes 0-31 of B, applied to the (0,1) and
            //   (2,3) partial results.  Afterwards lane row q holds tile q's fully merged result, i.e.
            //   lane (q, pix) owns row row0 + 16 q + pix = row0 + lane.  9 swaps per 64 rows.
            auto merge = [&](float x1, float y1, float x2, float y2, float xs, float ys, bool wide, float &o1,
                             float &o2, float &os) {
                uint2v r1, r2, rs;
                if (wide) {
                    r1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(x1), __float_as_uint(y1), false, false);
                    r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(x2), __float_as_uint(y2), false, false);
                    rs = __builtin_amdgcn_permlane32_swap(__float_as_uint(xs), __float_as_uint(ys), false, false);
                } else {
                    r1 = __builtin_amdgcn_permlane16_swap(__float_as_uint(x1), __float_as_uint(y1), false, false);
                    r2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(x2), __float_as_uint(y2), false, false);
                    rs = __builtin_amdgcn_permlane16_swap(__float_as_uint(xs), __float_as_uint(ys), false, false);
                }
                const float a = __uint_as_float(r1[0]), b = __uint_as_float(r1[1]);
                o1 = max_of(a, b);
                o2 = fmaxf(fmaxf(min_of(a, b), __uint_as_float(r2[0])), __uint_as_float(r2[1]));
                os = __uint_as_float(rs[0]) + __uint_as_float(rs[1]);
            };
            float p1, p2, ps, q1, q2, qs, a1, a2, s2;
            merge(tm1[0], tm1[1], tm2[0], tm2[1], ss[0], ss[1], false, p1, p2, ps);
            merge(tm1[2], tm1[3], tm2[2], tm2[3], ss[2], ss[3], false, q1, q2, qs);
            merge(p1, q1, p2, q2, ps, qs, true, a1, a2, s2);
            // tol = trel (|X| wn_max + wn_max^2 / 2) + tol_abs (|X| + wn_max) with |X| <= 1.001 sqrt(s2) (|Xh| <= |X| (1 + 2^-11) and
            // the sqrt's ulp), as ONE fused multiply-add on sqrt(s2): tol_a, tol_b below.  A row is unfit for the filter when
            // 1.001 sqrt(s2) >= x_limit or s2 is not finite: one unsigned compare of the bit pattern of s2 (a sum of squares: never
            // negative; an infinity or a NaN of either sign compares above every finite bound) against that of x_limit^2 scaled down.
            unsigned sbits = __float_as_uint(s2);
            asm("" : "+v"(sbits));  // opaque copy: keeps the test on the bits under finite-math
            const unsigned long long unfit_m = __builtin_amdgcn_uicmp(sbits, s2_limit_bits, 35 /* unsigned >= */) | force_m;
            const float tol = fmaf(__builtin_amdgcn_sqrtf(s2), FULL ? tol_a : tol_a_coarse, FULL ? tol_b : tol_b_coarse);
            amb_m = ~__builtin_amdgcn_fcmpf(a1 - a2, tol, 2 /* ordered > */) | unfit_m;
            my_m1 = a1;
        }

        // Both waves of a SIMD run this same stream, so MFMA bursts and VALU stretches would line up
        // and the two pipes would take turns instead of overlapping (measured: VALU-active + MFMA-busy
        // ~= 100 % of the runtime).  Ask the scheduler for a fine interleave inside each wave:
        // every MFMA is followed by VALU work that does not depend on it.
        if constexpr (MODE != 1 && SGB_VALU > 0) {
#pragma unroll
            for (int i = 0; i < kTilesPerIter * NB * 3; i++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, SGB_VALU, 0);  // VALU
            }
        }
        // lane (q, pix) owns row row0 + q*16 + pix == row0 + lane (FULL: the row in slot `lane` of the wave's queue)
        int64_t row0 = 0, row;
        unsigned long long own_m;
        if constexpr (QUEUED) {
            row = s1_q[lane < (int)full_rows ? lane : 0];
            own_m = full_rows >= 64u ? ~0ull : (1ull << full_rows) - 1ull;
        } else {
            // rows of a shifted last group that the previous group already covered are not listed again
            // (a row listed twice would be accumulated twice by the exact kernel)
            const int shift = g == ngroups - 1 ? last_shift : 0;
            row0 = (int64_t)g * 64 - shift;
            row = row0 + lane;
            own_m = ~0ull << shift;
        }
        amb_m &= own_m;
        const bool my_amb = __builtin_amdgcn_inverse_ballot_w64(amb_m);
        {
            // id (q, b, r) -> node: 16 b + 4 q + r, the last block's 4x4 (q, r) grid transposed (node_of_row)
            const unsigned id = __float_as_uint(my_m1) & node_mask;
            const unsigned wq = id >> 5, wb = (id >> 2) & 7u, wr = id & 3u;
            const unsigned real = wb == (unsigned)(NB - 1) ? 16u * wb + 4u * wr + wq : 16u * wb + 4u * wq + wr;
            // ACC: rows of the shifted last group that the group before it owns are left alone -- that group may
            // already have settled them exactly inside this launch, and a late provisional store would undo it
            // (plain filter: the rewrite is harmless, the exact kernel runs in a launch of its own afterwards)
            // (a row that waits for the full search keeps no provisional label: the search's store is the only one)
            // (nor does a row of a shifted last group that the group before it owns: that group's wave may have searched it in
            // full already)
            if (__builtin_amdgcn_inverse_ballot_w64(DEFER ? own_m & ~amb_m : own_m)) {
                if constexpr (QUEUED) labels[row] = (int)real + 1;
                else (labels + row0)[lane] = (int)real + 1;   // (scalar base + the lane's constant offset: no 64-bit vector address)
            }
            if constexpr (ACC) {
                // lane (q, pix) holds channels q*CPL.. of rows (t, pix), t = 0..3; their labels sit in lanes (t, pix).
                // Listed rows and rows a previous group already added go to the spare row k, clamped channel slots
                // (they re-read the row's last valid pair) too: straight-line code, the four label exchanges in flight
                // together.
                const unsigned mine = __builtin_amdgcn_inverse_ballot_w64(amb_m | ~own_m) ? (unsigned)k : real;
                // Round 5: the four labels travel through v_permlane16/32_swap (VALU) instead of four ds_bpermute, whose
                // results waited in the LDS queue behind the trip's own atomics: swapping a value with itself hands every lane
                // {even row's, odd row's} of its pair of lane rows, then {lower half's, upper half's} of the wave.
                unsigned lab[kTilesPerIter];
                {
                    const uint2v r16 = __builtin_amdgcn_permlane16_swap(mine, mine, false, false);   // rows 0, 1: {L0, L1}; rows 2, 3: {L2, L3}
                    const uint2v e32 = __builtin_amdgcn_permlane32_swap(r16[0], r16[0], false, false);   // {L0, L2} in every lane
                    const uint2v o32 = __builtin_amdgcn_permlane32_swap(r16[1], r16[1], false, false);   // {L1, L3}
                    lab[0] = e32[0];
                    lab[1] = o32[0];
                    lab[2] = e32[1];
                    lab[3] = o32[1];
                }
                if constexpr (FIX) {
                    // Round 6: every slot of a lane adds what the lane loaded for it to ITS word of the row's table row
                    // (acc_stride_fix): address = label x stride + lane base, the pairs in the offset field.  Listed rows and rows
                    // a previous group owns carry the spare label k.  The slot at channel c (FOLD: it exists) loaded zeros and so
                    // adds the bare bit pattern of the magic number: the row count, see where fx is made.
                    const unsigned cs8 = (unsigned)cs * 8u;
                    char *const lane_base = reinterpret_cast<char *>(lu) + (unsigned)(q * CPL) * 8u + (unsigned)(pix & (kCopies - 1)) * (unsigned)(table_words * 8);
                    auto table_row = [&](unsigned label) { return reinterpret_cast<unsigned long long *>(lane_base + __umul24(label, cs8)); };
                    auto slot_bits = [&](float v) { return (unsigned long long)__double_as_longlong((double)v + fx.magic); };
                    // (scan_trip was decided a trip ago, on that trip's last tile -- labels that agreed there agree next door -- so the
                    // branch below does not wait for a vector compare of this trip's labels; rows that come out of the wave's queue
                    // are not neighbours -- and the idle lanes of a short batch all carry the spare label: such a trip neither takes
                    // the decision nor makes the next one)
                    const bool scan_now = QUEUED ? false : scan_trip;
                    if constexpr (FOLD && PXSOM_ADD_SCAN && !QUEUED) {
                        const unsigned nx3 = (unsigned)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)lab[kTilesPerIter - 1], 0x101 /* row_shl:1 */, 0xf, 0xf, false);
                        // agreeing neighbour pairs of that tile's 60 (15 per lane row, the four lane rows alike)
                        scan_trip = __popcll(__ballot(nx3 == lab[kTilesPerIter - 1])) >= PXSOM_ADD_SCAN_MIN;
                    }
                    // Round 5: neighbouring rows that share their label (what images do and the synthetic FOVs do not: the 16 rows of
                    // one ds_add then hit the same words and the instruction is served lane after lane -- 0.30 -> 0.84 ms on rows in
                    // runs of 16 equal labels, profiles/r05/label_coherence.txt).  Where most neighbours agree (a wave-uniform count)
                    // the tile's rows are first summed along the row axis: inclusive prefix sums P of the fixed-point words over the 16
                    // lanes of a lane row (v_add_co / v_addc on row_shr operands, two instructions per step and word), and only the
                    // LAST row e of every run of equal labels touches the table: + P[e] to its own label, - P[e] to the label of the
                    // run that follows (whose own last row adds its P, which contains P[e]): the telescoped sums are the runs' sums,
                    // in the same modular 64-bit arithmetic as the plain adds, so the table ends bit-identical.  (The count word is a
                    // slot like any other.)
                    if constexpr (FOLD && PXSOM_ADD_SCAN)
                    if (__builtin_expect(scan_now, 0)) {   // (out of line: the plain adds below stay one block behind the search)
                        // Round 6: ALL 64 rows of the trip under one label (runs of 64 rows and longer: the inside of a region of
                        // an image) -- the four tiles' words are first added up inside the lane (a lane holds the same channel slots
                        // of rows pix, 16 + pix, 32 + pix, 48 + pix), ONE reduction over the 16 lanes of a lane row follows instead
                        // of four prefix sums, and the last lane of every lane row adds its six words: 36 + 48 vector instructions
                        // and 6 atomics per 64 rows instead of 192 and 24.  Same modular sums: the table ends bit-identical.
                        const unsigned lab_first = (unsigned)__builtin_amdgcn_readfirstlane((int)lab[0]);
                        const bool one_label = __ballot(lab[0] == lab_first && lab[1] == lab_first && lab[2] == lab_first && lab[3] == lab_first) == ~0ull;
                        if (one_label) {
                            unsigned lo[2 * NP], hi[2 * NP];
#pragma unroll
                            for (int p = 0; p < NP; p++) {
                                unsigned long long sx = slot_bits((float)keep[0][p].x), sy = slot_bits((float)keep[0][p].y);
#pragma unroll
                                for (int t = 1; t < kTilesPerIter; t++) {
                                    sx += slot_bits((float)keep[t][p].x);
                                    sy += slot_bits((float)keep[t][p].y);
                                }
                                lo[2 * p] = (unsigned)sx;
                                hi[2 * p] = (unsigned)(sx >> 32);
                                lo[2 * p + 1] = (unsigned)sy;
                                hi[2 * p + 1] = (unsigned)(sy >> 32);
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
                            if (pix == 15) {   // (the inclusive prefix sum of the last lane of a lane row is the row's total)
                                unsigned long long *const one_row = table_row(lab_first);
#pragma unroll
                                for (int j = 0; j < 2 * NP; j++)
                                    __hip_atomic_fetch_add(one_row + j, ((unsigned long long)hi[j] << 32) | lo[j], __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_WORKGROUP);
                            }
                        } else
#pragma unroll
                        for (int t = 0; t < kTilesPerIter; t++) {
                            const unsigned nxt = (unsigned)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)lab[t], 0x101 /* row_shl:1 */, 0xf, 0xf, false);
                            const bool last = pix == 15;
                            const bool ends = last || nxt != lab[t];
                            unsigned lo[2 * NP], hi[2 * NP];
#pragma unroll
                            for (int p = 0; p < NP; p++) {
                                const unsigned long long bx = slot_bits((float)keep[t][p].x), by = slot_bits((float)keep[t][p].y);
                                lo[2 * p] = (unsigned)bx;
                                hi[2 * p] = (unsigned)(bx >> 32);
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
                                unsigned long long *const mine_row = table_row(lab[t]);
                                unsigned long long *const next_row = table_row(nxt < (unsigned)k ? nxt : (unsigned)k);
#pragma unroll
                                for (int j = 0; j < 2 * NP; j++) {
                                    const unsigned long long pj = ((unsigned long long)hi[j] << 32) | lo[j];
                                    __hip_atomic_fetch_add(mine_row + j, pj, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                    if (!last) __hip_atomic_fetch_add(next_row + j, 0ull - pj, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                }
                            }
                        }
                    }
                    if (__builtin_expect(!scan_now, 1))
#pragma unroll
                    for (int t = 0; t < kTilesPerIter; t++) {
                        unsigned long long *const row_words = table_row(lab[t]);
#pragma unroll
                        for (int p = 0; p < NP; p++) {
                            __hip_atomic_fetch_add(row_words + 2 * p, slot_bits((float)keep[t][p].x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            __hip_atomic_fetch_add(row_words + 2 * p + 1, slot_bits((float)keep[t][p].y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                        if constexpr (!FOLD)   // (c == 4 CPL: no slot past the row's end -- the first lane group counts the row)
                            if (q == 0) __hip_atomic_fetch_add(row_words + c, fx.mbits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                } else {
                    // binary64 tables (the training steps' statistics): a label's row or the spare one per pair, the count in the
                    // idle lanes' first add where there are any (c < 4 CPL) and by an instruction of its own otherwise
                    const bool fold = c < 4 * CPL;                      // (wave-uniform)
                    const bool cnt_lane = fold && q == 3;
                    const unsigned cnt_base = (unsigned)(k + 1) * (unsigned)cs;
#pragma unroll
                    for (int t = 0; t < kTilesPerIter; t++) {
                        const unsigned base = __umul24(lab[t], (unsigned)cs), spare = __umul24((unsigned)k, (unsigned)cs);   // (v_mul_lo_u32 runs at quarter rate)
#pragma unroll
                        for (int p = 0; p < NP; p++) {
                            const bool own = q * CPL + 2 * p <= c - 2;
                            unsigned idx = (own ? base : spare) + (unsigned)(own ? q * CPL + 2 * p : 0);
                            const bool counts_here = p == NP - 1 && cnt_lane;
                            const unsigned idx0 = counts_here ? cnt_base + lab[t] : idx;
                            __hip_atomic_fetch_add(ls + idx0, counts_here ? 1.0 : (double)keep[t][p].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            __hip_atomic_fetch_add(ls + idx + 1, (double)keep[t][p].y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                        if (!fold && q == 0)
                            __hip_atomic_fetch_add(ls + (size_t)(k + 1) * cs + lab[t], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
        }
        const unsigned long long mask = amb_m;
        if constexpr (DEFER) {
            if (mask) {   // into the wave's queue, behind what it holds (room for two trips' rows is checked by the group loop)
                if (my_amb) s1_q[s1_n + (unsigned)__popcll(mask & ((1ull << lane) - 1ull))] = row;
                s1_n += (unsigned)__popcll(mask);
            }
        } else if (mask) {
            if constexpr (ACC) {
                unsigned base = 0;
                if (lane == 0) base = atomicAdd(amb_n, (unsigned)__popcll(mask));
                base = __shfl(base, 0);
                const unsigned pos = base + (unsigned)__popcll(mask & ((1ull << lane) - 1ull));
                if (my_amb && pos < kAmbQueue) amb_q[pos] = row;
                // queue full (a launch over many groups): the rows that did not fit are settled on the spot
                unsigned long long late = __ballot(my_amb && pos >= kAmbQueue);
                while (late) {
                    const int src = __builtin_ctzll(late);
                    late &= late - 1;
                    const int64_t rsrc = QUEUED ? s1_q[src] : row0 + src;
                    exact_row_accumulate<T, FIX>(x, rsrc, c, ldx, wx, k, labels, ls, lane, &fx, stats, cs, wsj, wsn, FIX ? c : -1, FIX ? fx.mbits : 1ull);
                }
            } else {
                unsigned base = 0;
                if (lane == 0) base = atomicAdd(&hdr->amb_count, (unsigned)__popcll(mask));
                base = __shfl(base, 0);
                if (my_amb) amb_list[base + __popcll(mask & ((1ull << lane) - 1ull))] = (unsigned)row;
            }
        }
    };
    // the rows stage 1 left in the wave's queue, 64 at a time (`all`: the rest too), each batch gathered into a row set of
    // its own and searched in full
    // (every place the queue is emptied at comes behind a trip that read rows_b -- or behind the last trip of all: rows_b is free)
    RowSet &rows_q = rows_b;
    auto drain = [&](bool all) {
        while (s1_n >= 64u || (all && s1_n > 0u)) {
            const unsigned cnt = s1_n < 64u ? s1_n : 64u;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int t = 0; t < kTilesPerIter; t++) {
                const int slot = t * 16 + pix;
                const T *rp = x + s1_q[slot < (int)cnt ? slot : 0] * ldx;
#pragma unroll
                for (int p = 0; p < NP; p++) {
                    int ch = q * CPL + 2 * p;
                    if (ch > c - 2) ch = c - 2;
                    rows_q[t][p] = *reinterpret_cast<const P2 *>(rp + ch);
                    if constexpr (ACC && FIX)   // (a pair past the row's end holds zeros, as the group loads deliver it)
                        if (q * CPL + 2 * p > c - 2) rows_q[t][p] = P2{};
                }
            }
            trip(rows_q, rows_q, std::integral_constant<int, 1>{}, cnt);
            __builtin_amdgcn_wave_barrier();
            // what came in behind the batch moves to the front, 64 entries at a time in ascending order (up to 192 entries:
            // the queue holds two trips' worth beyond the level the group loop stops at)
            for (unsigned i = (unsigned)lane; i + 64u < s1_n; i += 64u) {
                const int64_t moved = s1_q[64u + i];
                __builtin_amdgcn_wave_barrier();
                s1_q[i] = moved;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            s1_n -= cnt;
        }
    };
    // (the full search stays OUT of the loop that streams the groups: that loop runs until the wave's queue could overflow within
    // two more trips -- on ordinary data never -- and the queue is emptied between two runs of it.  A fall-back that searches a
    // wave's groups in full straight away once most of its rows fail stage 1 -- crowded nodes -- was measured on the same box:
    // it takes a codebook with node pairs 1e-2 apart from 1.65 to 1.20 ms and costs the ordinary case 0.260 -> 0.270 ms for
    // being there; not kept.)
    // (Round 6, timing builds -DPXSOM_TIMING_ABL: the full search of the queued rows behind a wave's last group costs 9 us of the
    // 0.22 ms launch -- every wave of the chip runs it at the same moment with the gather's trip to HBM exposed --, the flush 5 us, the
    // listed rows 2 us.  Emptying the queue two trips early and sending the last trips' unsure rows straight to the exact path was
    // measured SLOWER: 0.2256 - 0.2311 against 0.2199 - 0.2329 ms, and 10 - 14 % slower on crowded codebooks; not kept.)
    while (g < ngroups) {
        while (g < ngroups && s1_n <= kS1Queue - 128u) {
            trip(rows_a, rows_b, std::integral_constant<int, 0>{}, 0u);
            g += nwaves;
            if (g >= ngroups) break;
            trip(rows_b, rows_a, std::integral_constant<int, 0>{}, 0u);
            g += nwaves;
        }
        if constexpr (MODE != 1 && !(PXSOM_TIMING_ABL & 4)) drain(true);   // (the one place the full search is instantiated)
        else s1_n = 0u;
    }
    if constexpr (ACC) {
        __syncthreads();   // every wave is through its groups: the queue is complete
        const unsigned queued = *amb_n < kAmbQueue ? *amb_n : kAmbQueue;   // rows past the end were settled at once
        for (unsigned i = threadIdx.x >> 6; i < queued; i += kWaves)
            if (!(PXSOM_TIMING_ABL & 2))
            exact_row_accumulate<T, FIX>(x, amb_q[i], c, ldx, wx, k, labels, ls, lane, &fx, stats, cs, wsj, wsn, FIX ? c : -1, FIX ? fx.mbits : 1ull);
        __syncthreads();
        if constexpr (FIX) {
            int node = (int)threadIdx.x / c, j = (int)threadIdx.x - node * c;   // element e <-> (node, channel), no division per element
            const int dnode = kThreads / c, dj = kThreads % c;
            // word c of a table row holds n B modulo 2^64 for the n rows of the label (B: the magic number's odd bit pattern) -- the
            // very amount those rows put on top of their values in every channel word
            auto word_sum = [&](size_t at) {   // a table word over the copies (modular: the order does not matter)
                unsigned long long t = lu[at];
#pragma unroll
                for (int r = 1; r < kCopies; r++) t += lu[(size_t)r * table_words + at];
                return t;
            };
            for (int e = threadIdx.x; e < k * c; e += kThreads) {
                const unsigned long long nb = word_sum((size_t)node * cs + c);
                if (nb) {
                    const long long units = (long long)(word_sum((size_t)node * cs + j) - nb);
                    if (units && !(PXSOM_TIMING_ABL & 1)) __hip_atomic_fetch_add(stats + e, (double)units * fx.unit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                node += dnode;
                j += dj;
                if (j >= c) {
                    j -= c;
                    node++;
                }
            }
            const unsigned long long binv = inverse_mod_2_64(fx.mbits);
            for (int e = threadIdx.x; e < k; e += kThreads) {
                const unsigned long long cnt = word_sum((size_t)e * cs + c) * binv;
                if (cnt) __hip_atomic_fetch_add(stats + (size_t)k * c + e, (double)cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // Round 6: the last workgroup through its flush turns the statistics into the caller's tables -- the finishing launch
            // behind this kernel is gone.  The statistics only ever see device-scope atomics, which are performed at the memory
            // side: a wave whose vmcnt has drained knows its adds are in, and the last workgroup reads them with device-scope
            // atomic loads.  NO release fence: at agent scope that is a write-back of the XCD's whole L2 -- 40 MB of freshly
            // written labels -- by every one of 512 workgroups (measured: kernel 0.22 -> 0.31 ms).
            if (fin.ticket) {
                __shared__ int s_last_wg;
                __builtin_amdgcn_s_waitcnt(0);
                __syncthreads();
                if (threadIdx.x == 0) {
                    const unsigned t = __hip_atomic_fetch_add(fin.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    s_last_wg = t == gridDim.x - 1u;
                }
                __syncthreads();
                if (s_last_wg) {
                    int node = (int)threadIdx.x / c, j = (int)threadIdx.x - node * c;
                    for (int e = threadIdx.x; e < k * c; e += kThreads) {
                        const double s = __hip_atomic_load(stats + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (fin.overwrite) {
                            const double cnt = __hip_atomic_load(stats + (size_t)k * c + node, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            fin.sums[e] = s;
                            if (fin.means) fin.means[e] = s / (cnt > 0.0 ? cnt : 1.0);
                        } else {
                            fin.sums[e] += s;
                        }
                        node += dnode;
                        j += dj;
                        if (j >= c) {
                            j -= c;
                            node++;
                        }
                    }
                    for (int e = threadIdx.x; e < k; e += kThreads) {
                        const long long cnt = (long long)__hip_atomic_load(stats + (size_t)k * c + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (fin.overwrite) fin.counts[e] = cnt;
                        else fin.counts[e] += cnt;
                    }
                    // ... and leaves the statistics region zero for the next call (include/pxsom.h PXSOM_TABLES_SCRATCH_CLEAN)
                    __syncthreads();
                    for (int e = threadIdx.x; e < k * c + k; e += kThreads)
                        __hip_atomic_store(stats + e, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (threadIdx.x == 0) __hip_atomic_store(fin.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        } else {
            int node = (int)threadIdx.x / c, j = (int)threadIdx.x - node * c;
            const int dnode = kThreads / c, dj = kThreads % c;
            for (int e = threadIdx.x; e < k * c; e += kThreads) {
                const double v = ls[(size_t)node * cs + j];
                if (v != 0.0) __hip_atomic_fetch_add(stats + e, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                node += dnode;
                j += dj;
                if (j >= c) {
                    j -= c;
                    node++;
                }
            }
            for (int e = threadIdx.x; e < k; e += kThreads) {
                const double v = ls[(size_t)(k + 1) * cs + e];   // counts sit behind the spare row
                if (v != 0.0) __hip_atomic_fetch_add(stats + (size_t)k * c + e, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

}  // namespace
}  // namespace pxsom_bmu
