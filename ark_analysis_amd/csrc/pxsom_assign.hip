// pxsom_assign.hip -- K7: best-matching-unit search on gfx950 (replaces pyFlowSOM.map_data_to_nodes,
// reference call site /root/reference/src/ark/phenotyping/cluster_helpers.py:150-157).
//
// Three launches per call, all on the caller's stream:
//   1. bmu_prep_kernel    (1 workgroup)  codebook f64 [K,C] -> MFMA A-fragments (fp16 hi/lo split,
//                          power-of-two scaled), per-node bias -0.5*|W_k|^2, error-bound constants.
//   2. bmu_filter_kernel  (persistent, 2 workgroups/CU) streams the pixel matrix once.  Scores
//                          s_k = X.W_k - 0.5|W_k|^2 (argmax_k s_k == argmin_k |x - w_k|) come from
//                          v_mfma_f32_16x16x32_f16 with the 3-term split Xh*Wh + Xl*Wh + Xh*Wl;
//                          rows = MFMA "N" (one pixel per lane&15), nodes = MFMA "M", so a
//                          pixel's scores sit in 4 lanes x 4 regs per node block and the top-2
//                          reduction is register-local (node index packed into the low mantissa
//                          bits) + two cross-lane merges.  A row whose best two scores are closer
//                          than a rigorous bound on the filter's error is appended to a list.
//   3. bmu_exact_kernel   re-evaluates listed rows exactly as the oracle does (binary64, j ascending,
//                          one rounding per op, sqrt, first strict minimum).  The label of every
//                          row is therefore bit-identical to the reference algorithm's,
//                          independent of the filter's precision.
// HBM traffic per pixel: C*sizeof(T) read + 4 written (DESIGN.md "K7").
#include <cfloat>
#include <cmath>
#include <cstdlib>

#include "pxsom_assign.h"
#include "pxsom_prep.h"
#include "pxsom_wave.h"

using namespace pxsom_bmu;

namespace {

// Lists at least this long go to bmu_exact_screened_kernel (a wave per 64 rows: throughput), shorter ones to
// bmu_exact_kernel (a wave per row or four: latency).  A wave of the former walks all K nodes at ~1 us each whatever
// the list's length; the latter settles a row in about 0.4 ps x K x C on the whole chip: they meet near 2.25e6 / C
// rows (measured: 100 K rows at C = 22, 22 K at C = 100).
// pxsom_assign_ex(..., PXSOM_ASSIGN_SCREEN_ALL_LISTS) sends every list there (the tests' route to that kernel on short lists).
// Codebooks whose binary32 copy the screened kernel stages in LDS (12 .. 64 KB): its lane groups share the nodes of a short
// batch, and the crossover drops to 8 K rows (measured on config 4, 100 x 100: pass 2.68 / 2.70 / 2.64 / 2.91 ms with the
// crossover at 22.5 K / 2 K / 8 K / 512 rows).
static unsigned screen_min_rows(int c, bool w32_in_lds, bool all_lists)
{
    if (all_lists) return 1u;
    const unsigned by_width = (unsigned)(2250000 / (c > 0 ? c : 1));
    return w32_in_lds ? std::min(by_width, 8192u) : by_width;
}

// NT threads in ONE workgroup: 256, or 1024 for codebooks of more than 128 nodes (every phase is a loop over
// nodes or fragments: four times the threads, a quarter of the trips)
template <int NT>
__global__ __launch_bounds__(NT) void bmu_prep_kernel(const double *__restrict__ w, int k, int c,
                                                       AssignHdr *hdr, half8 *wfrag, f32x4 *bias,
                                                       int nb, int nch, int cpl, int idx_bits,
                                                       int node_bits, int stage, double *zero_ptr,
                                                       int zero_count, double *wt_out, float *w32_out, int cp32, int npk,
                                                       int center)
{
    PXSOM_PHASE_ANY(0);
    // fused batch accumulation: the statistics buffer is cleared here instead of by a memset node
    for (int e = threadIdx.x; e < zero_count; e += NT) zero_ptr[e] = 0.0;
    extern __shared__ __attribute__((aligned(16))) char prep_smem[];
    const int tid = threadIdx.x;
    // small codebooks are staged in LDS with one coalesced sweep (8 loads in flight per thread); every
    // later read is an LDS read
    double *sw = reinterpret_cast<double *>(prep_smem);
    if (stage) {
        for (int e0 = tid; e0 < k * c; e0 += 8 * NT) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = w[e0 + u * NT < k * c ? e0 + u * NT : 0];
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (e0 + u * NT < k * c) sw[e0 + u * NT] = v[u];
        }
    }
    __syncthreads();
    PXSOM_PHASE_ANY(1);
    // two calls, not one with a selected pointer: each inlined copy then knows its address space (ds_read for
    // the staged codebook instead of flat loads)
    if (stage) prep_body<NT>(sw, k, c, hdr, wfrag, bias, nb, nch, cpl, idx_bits, node_bits, wt_out, w32_out, cp32, npk, center != 0);
    else prep_body<NT>(w, k, c, hdr, wfrag, bias, nb, nch, cpl, idx_bits, node_bits, wt_out, w32_out, cp32, npk, center != 0);
}


// ------------------------------------------------------------------------------------------------
// 3. exact path: one wave per listed row; lane <-> nodes lane, lane+64, ...
//    binary64, no contraction, j ascending, sqrt, first strict minimum (FlowSOM C_mapDataToCodes).
// ------------------------------------------------------------------------------------------------
#pragma clang fp contract(off)
// One batch of RB listed rows on one wave.  RB = 4: the codebook element read from LDS is shared by four
// rows and their independent binary64 chains hide each other's latency (long lists); RB = 1: one row per
// wave, so a short list (a training mini-batch lists a handful of rows) spreads over many waves and
// finishes in one row's latency.
template <typename T, int RB>
struct ExactRows {
    int64_t rows[RB];
    unsigned x_lo[RB][2], x_hi[RB][2];

    // row numbers, then the rows themselves (lane j holds channels j and j+64): two dependent HBM/L2
    // round trips, issued for the NEXT batch before the current one is evaluated
    __device__ __forceinline__ void load(const T *__restrict__ x, int c, int64_t ldx,
                                         const unsigned *__restrict__ amb_list, unsigned e0, unsigned count, int lane,
                                         const pxsom::RowView &rv)
    {
#pragma unroll
        for (int u = 0; u < RB; u++) {
            const unsigned e = e0 + u < count ? e0 + u : count - 1;  // surplus slots redo the last row
            rows[u] = amb_list[e];
        }
#pragma unroll
        for (int u = 0; u < RB; u++) {
            const T *rp = x + rv.offset(rows[u], ldx);
            const double xa = (double)rp[lane < c ? lane : 0];
            const double xb = (double)rp[lane + 64 < c ? lane + 64 : 0];
            x_lo[u][0] = (unsigned)__double_as_longlong(xa);
            x_hi[u][0] = (unsigned)(__double_as_longlong(xa) >> 32);
            x_lo[u][1] = (unsigned)__double_as_longlong(xb);
            x_hi[u][1] = (unsigned)(__double_as_longlong(xb) >> 32);
        }
    }
};

// NS node slots per sweep (lane <-> nodes base + lane + 64 s), JB channels of codebook values requested
// together.  Long lists (RB = 4) are throughput-bound: NS = 2, JB = 8.  A short list is one row per wave and pure
// latency; against a big codebook read from L2 its sweeps collapse into one with NS = 8 (512 nodes), JB = 4.
template <typename T, int RB, int NS, int JB>
__device__ __forceinline__ void exact_rows_loop(const T *__restrict__ x, int c, int64_t ldx,
                                                const double *__restrict__ w, const double *wt, int k,
                                                unsigned count, const unsigned *__restrict__ amb_list,
                                                int32_t *__restrict__ labels, int use_lds,
                                                unsigned wave, unsigned nwaves,
                                                int lane, ExactRows<T, RB> &cur, const pxsom::RowView &rv)
{
    for (unsigned e0 = wave * RB; e0 < count; e0 += nwaves * RB) {
        ExactRows<T, RB> nxt;
        const unsigned en = e0 + nwaves * RB;
        if (en < count) nxt.load(x, c, ldx, amb_list, en, count, lane, rv);
        double best[RB];
        int bestk[RB];
#pragma unroll
        for (int u = 0; u < RB; u++) {
            best[u] = DBL_MAX;
            bestk[u] = 0x7fffffff;
        }
        for (int base = 0; base < k; base += 64 * NS) {
            int cn[NS];   // clamped node of each slot
            double d[NS][RB];
#pragma unroll
            for (int s = 0; s < NS; s++) {
                const int nd = base + 64 * s + lane;
                cn[s] = nd < k ? nd : k - 1;
#pragma unroll
                for (int u = 0; u < RB; u++) d[s][u] = 0.0;
            }
            // LDS copy, or (big codebooks) the transposed copy prep left in the workspace: either way lanes read
            // consecutive nodes of channel j.  JB channels' values of every slot are requested together: one at
            // a time, a short list (a training mini-batch: one row per wave) pays the L2 latency per channel.
            for (int j0 = 0; j0 < c; j0 += JB) {
                double wv[NS][JB];
#pragma unroll
                for (int i = 0; i < JB; i++) {
                    const int j = j0 + i < c ? j0 + i : c - 1;
#pragma unroll
                    for (int s = 0; s < NS; s++) wv[s][i] = wt[(size_t)j * k + cn[s]];
                }
#pragma unroll
                for (int i = 0; i < JB; i++) {
                    const int j = j0 + i;
                    if (j < c) {   // uniform
                        const int h = j >> 6, jj = j & 63;
#pragma unroll
                        for (int u = 0; u < RB; u++) {
                            const unsigned lo = __builtin_amdgcn_readlane(h ? cur.x_lo[u][1] : cur.x_lo[u][0], jj);
                            const unsigned hi = __builtin_amdgcn_readlane(h ? cur.x_hi[u][1] : cur.x_hi[u][0], jj);
                            const double xj = __longlong_as_double(((long long)hi << 32) | lo);
#pragma unroll
                            for (int s = 0; s < NS; s++) {
                                const double t = xj - wv[s][i];
                                d[s][u] += t * t;
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int s = 0; s < NS; s++) {   // ascending node order within the lane: first strict minimum
                const int nd = base + 64 * s + lane;
#pragma unroll
                for (int u = 0; u < RB; u++) {
                    const double sd = sqrt(d[s][u]);
                    if (nd < k && sd < best[u]) {
                        best[u] = sd;
                        bestk[u] = nd;
                    }
                }
            }
        }
        PXSOM_PHASE(11);
#pragma unroll
        for (int u = 0; u < RB; u++) {
            // first strict minimum over the wave: smallest distance (NaN distances never replaced
            // DBL_MAX), then the smallest node index among the lanes that hold it -- DPP/permlane only
            const double smin = pxsom::wave_min_f64(best[u]);
            const unsigned cand = best[u] == smin ? (unsigned)bestk[u] : 0xffffffffu;
            const int win = (int)pxsom::wave_min_u32(cand);  // 0x7fffffff: no finite distance (NaN row)
            if (lane == 0) labels[cur.rows[u]] = win == 0x7fffffff ? 0 : win + 1;
        }
        PXSOM_PHASE(12);
        if (en < count) cur = nxt;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bmu_exact_kernel(const T *__restrict__ x, int c, int64_t ldx,
                                                        const double *__restrict__ w, int k,
                                                        const AssignHdr *hdr,
                                                        const unsigned *__restrict__ amb_list,
                                                        int32_t *__restrict__ labels, int use_lds,
                                                        const double *__restrict__ wt_global, unsigned screened_from, pxsom::RowView rv)
{
    if (hdr->amb_count >= screened_from) return;   // long lists: bmu_exact_screened_kernel (launched beside this one)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    // [c][k] transposed codebook: staged in LDS, or (too big for LDS) the copy prep wrote to the workspace
    const double *wt = use_lds ? reinterpret_cast<const double *>(smem_raw) : wt_global;
    PXSOM_PHASE(8);
    const unsigned count = hdr->amb_count;
    const int lane = threadIdx.x & 63;
    const unsigned wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    const bool wide = count > nwaves;  // uniform over the grid: more rows than waves -> 4 rows per wave
    if (blockIdx.x * 4u * (wide ? 4u : 1u) >= count) return;  // uniform per workgroup: nothing listed for it
    PXSOM_PHASE(9);
    // the first batch's two dependent round trips overlap the codebook staging below
    ExactRows<T, 4> cur4;
    ExactRows<T, 1> cur1;
    if (wide) {
        if (wave * 4u < count) cur4.load(x, c, ldx, amb_list, wave * 4u, count, lane, rv);
    } else {
        if (wave < count) cur1.load(x, c, ldx, amb_list, wave, count, lane, rv);
    }
    if (use_lds) {
        // 8 independent loads in flight per thread (a one-load-per-trip loop pays the L2 latency per trip)
        for (int e0 = threadIdx.x; e0 < k * c; e0 += 8 * 256) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = w[e0 + u * 256 < k * c ? e0 + u * 256 : 0];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int e = e0 + u * 256;
                if (e < k * c) {
                    const int node = e / c, j = e - node * c;
                    reinterpret_cast<double *>(smem_raw)[(size_t)j * k + node] = v[u];
                }
            }
        }
        __syncthreads();
    }
    PXSOM_PHASE(10);
    if (wide)
        exact_rows_loop<T, 4, 2, 8>(x, c, ldx, w, wt, k, count, amb_list, labels, use_lds, wave, nwaves, lane, cur4, rv);
    else if (!use_lds && k > 128)
        exact_rows_loop<T, 1, 8, 4>(x, c, ldx, w, wt, k, count, amb_list, labels, use_lds, wave, nwaves, lane, cur1, rv);
    else
        exact_rows_loop<T, 1, 2, 8>(x, c, ldx, w, wt, k, count, amb_list, labels, use_lds, wave, nwaves, lane, cur1, rv);
}

// ------------------------------------------------------------------------------------------------
// 3b. exact path for LONG lists (screen_min_rows(): crowded codebooks, discrete data, degenerate early codebooks).
//     The kernel above spends K*C binary64 operations per listed row on nodes that are nowhere near the winner.
//     Here lanes <-> listed rows (64 per wave) and the nodes are walked uniformly:
//       1. reference distance D_ref: the oracle's arithmetic for the node the filter proposed for the row -- the
//          distance of SOME node, hence an upper bound of the minimum;
//       2. screening pass in packed binary32 over all K nodes (codebook values from the binary32 copy prep left
//          in the workspace, read through the scalar cache: no vector memory traffic, no cross-lane traffic;
//          one v_pk_add + one v_pk_fma per channel pair and 64 rows): node k stays a candidate unless its
//          binary32 squared distance exceeds T = ((D_ref (1 + 1e-12) + delta) ^ 2) (1 + eta), where
//          delta = 2^-24 (|x| + max|w|) bounds what rounding x and w to binary32 can move a distance (triangle
//          inequality) and eta = (C + 4) 2^-24 the rounding of the binary32 sum of squares (eight chains of C/8
//          fused multiply-adds); a node the oracle could pick has D_k <= D_ref and therefore passes.  Rows or
//          codebooks outside binary32's comfortable range keep every node (T = inf);
//       3. the surviving (row, node) pairs -- one or two per row as a rule -- are queued in LDS and evaluated lane
//          per pair exactly as the oracle does (binary64, j ascending, no contraction, sqrt); the row keeps the
//          smallest distance and, among equal ones, the smallest node: the oracle's first strict minimum.
// ------------------------------------------------------------------------------------------------
constexpr int kPairCap = 512;
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// WLDS: the binary32 codebook copy is staged in LDS and read as broadcasts.  The scalar cache (16 KB) serves a small table
// at a few cycles per load; a table that does not fit it (config 4: 100 x 104 floats = 42 KB) turns every 64-byte scalar
// load into an L2 round trip the wave waits for -- seven per node, 465 us for a batch of 64 rows.
template <typename T, int CB, int NU, bool WLDS>   // c <= 8 * CB = Layout::cp32; NU nodes' codebook rows requested together
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((CB <= 3 || (sizeof(T) <= 4 && CB <= 5)) ? 4 : 2)))
void bmu_exact_screened_kernel(const T *__restrict__ x, int c, int64_t ldx,
                                                                 const double *__restrict__ w,
                                                                 const float *__restrict__ w32, int k,
                                                                 const AssignHdr *hdr,
                                                                 const unsigned *__restrict__ amb_list,
                                                                 int32_t *labels, unsigned min_rows, pxsom::RowView rv)
{
    const unsigned count = hdr->amb_count;
    if (count < min_rows) return;
    __shared__ unsigned long long s_bestd[4][64];
    __shared__ int s_bestk[4][64];
    __shared__ unsigned s_pairs[4][kPairCap];
    __shared__ long long s_row[4][64];
    constexpr bool kNarrow = sizeof(T) <= 4;   // binary16 / binary32 rows: a NaN / Inf shows in the binary32 copy
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (blockIdx.x * 16u >= count) return;   // uniform per workgroup: no batch for it even at four rows per wave
    extern __shared__ __attribute__((aligned(16))) float s_w32[];
    if constexpr (WLDS) {
        const int words4 = k * (8 * CB) / 4;
        const f32x4 *src = reinterpret_cast<const f32x4 *>(w32);
        for (int e0 = threadIdx.x; e0 < words4; e0 += 4 * 256) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = src[e0 + u * 256 < words4 ? e0 + u * 256 : 0];
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (e0 + u * 256 < words4) reinterpret_cast<f32x4 *>(s_w32)[e0 + u * 256] = v[u];
        }
        __syncthreads();
    }
    const double u24 = 0x1p-24;
    const bool blind = hdr->force_exact != 0 || !(hdr->scale > 0.f);
    const double wn = blind ? 0.0 : (double)hdr->wn_raw;   // (uncentred: the bound is on rounding x and w themselves)
    const double eta = (double)(c + 4) * u24;
    const unsigned long long kInitD = (unsigned long long)__double_as_longlong(DBL_MAX);

    // A wave takes 64 / G listed rows at a time and G lane groups share the nodes among them (group g: nodes
    // [g * kper, (g + 1) * kper)): a batch costs 1 / G of the node walk, so a list too short to give every wave of the
    // launch 64 rows still spreads over all of them, and a long one ends in a short last round.  G is the power of two
    // that minimises rounds x (1 / G + fixed part); the LDS copy is what lets the groups read different nodes (the
    // scalar path reads one node for the whole wave: G = 1).
    const unsigned nwaves = gridDim.x * 4;
    int G = 1;
    if constexpr (WLDS) {
        float best = 3.0e38f;
        for (int g = 1; g <= 16; g *= 2) {
            const unsigned nb = (count * (unsigned)g + 63u) / 64u;
            const float cost = (float)((nb + nwaves - 1) / nwaves) * (1.0f / (float)g + 0.24f);
            if (cost < best) { best = cost; G = g; }
        }
    }
    const int rpw = 64 / G, kper = (k + G - 1) / G;
    const int grp = lane / rpw, slot = lane % rpw;
    const int node_lo = grp * kper, node_hi = min(k, node_lo + kper);
    const unsigned nbatches_g = (count + (unsigned)rpw - 1) / (unsigned)rpw;

    for (unsigned bt = blockIdx.x * 4 + wv; bt < nbatches_g; bt += nwaves) {
        const unsigned e = bt * (unsigned)rpw + (unsigned)slot;
        const bool valid = e < count;
        const int64_t row = amb_list[valid ? e : count - 1];
        const T *rp = x + rv.offset(row, ldx);
        float xv[8 * CB];
#pragma unroll
        for (int j = 0; j < 8 * CB; j++) {
            const float v = (float)rp[j < c ? j : c - 1];
            xv[j] = j < c ? v : 0.f;
        }
        if (grp == 0) s_row[wv][slot] = row;
        float n2 = 0.f;
        bool finite_x = true;
#pragma unroll
        for (int j = 0; j < 8 * CB; j++) {
            n2 = fmaf(xv[j], xv[j], n2);
            finite_x &= fabsf(xv[j]) <= FLT_MAX;
        }
        int ref = labels[row] - 1;
        if ((unsigned)ref >= (unsigned)k) ref = 0;
        wave_lds_sync();
        // the oracle's distance of the listed row in slot `rl` to one node.  binary16 / binary32 rows: xv IS the row (the
        // conversion to binary32 is exact), so the value comes out of the registers of lane rl (group 0 holds slot rl in
        // lane rl; ds_bpermute: no memory traffic); binary64 rows are re-read (L2)
        auto oracle_distance = [&](int node, int rl, int64_t r) __attribute__((always_inline)) {
#pragma clang fp contract(off)
            const double *wk = w + (size_t)node * c;
            double acc = 0.0;
            if constexpr (kNarrow) {
                // eight codebook values requested together; a channel past c adds an exact +0.0 (no branch per channel:
                // the compiler turns one into a load -> wait -> add chain of a hundred round trips)
#pragma unroll
                for (int j0 = 0; j0 < 8 * CB; j0 += 8) {
                    double wv8[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) wv8[u] = wk[j0 + u < c ? j0 + u : 0];
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const float xj = __shfl(xv[j0 + u], rl, 64);
                        double t = (double)xj - wv8[u];
                        t = j0 + u < c ? t : 0.0;
                        acc += t * t;
                    }
                }
            } else {
                const T *rq = x + rv.offset(r, ldx);
#pragma unroll(CB >= 16 ? 4 : 8)
                for (int j = 0; j < c; j++) {
                    const double t = (double)rq[j] - wk[j];
                    acc += t * t;
                }
            }
            return sqrt(acc);
        };
        const double dref = oracle_distance(ref, lane, row);   // (every group for its own copy of the row: same value)
        // the reference node's distance is the row's first candidate (the pairs below leave that node out)
        if (grp == 0) {
            const bool seeded = dref < DBL_MAX;   // NaN / Inf never replace the oracle's DBL_MAX
            s_bestd[wv][slot] = seeded ? (unsigned long long)__double_as_longlong(dref) : kInitD;
            s_bestk[wv][slot] = seeded ? ref : 0x7fffffff;
        }
        wave_lds_sync();
        // rows with a NaN / Inf keep label 0 in the oracle (no distance is ever below DBL_MAX): nothing to evaluate
        const bool skip = !valid || (kNarrow && !finite_x);
        float thr = __builtin_inff();
        if (!blind && n2 < 1.0e36f) {
            const double delta = u24 * (sqrt((double)n2) + wn) * 1.001 + 1.0e-30;
            const double tt = dref * (1.0 + 1.0e-12) + delta;
            const double t64 = tt * tt * (1.0 + eta) * (1.0 + 0x1p-20);
            if (t64 < 3.0e38) thr = fmaxf((float)t64, 1.2e-38f);
        }

        unsigned npairs = 0;
        auto flush = [&]() __attribute__((always_inline)) {
            for (unsigned p0 = 0; p0 < npairs; p0 += 64) {
                const unsigned p = p0 + lane;
                const bool on = p < npairs;
                const unsigned pr = s_pairs[wv][on ? p : 0];
                const int rl = (int)(pr >> 16), node = (int)(pr & 0xffffu);
                const double dist = oracle_distance(node, rl, kNarrow ? 0 : s_row[wv][rl]);
                const bool ok = on && dist < DBL_MAX;          // NaN / Inf never replace the oracle's DBL_MAX
                const unsigned long long bits = (unsigned long long)__double_as_longlong(dist);
                const unsigned long long before = s_bestd[wv][rl];
                wave_lds_sync();
                if (ok) atomicMin(&s_bestd[wv][rl], bits);
                wave_lds_sync();
                const unsigned long long now = s_bestd[wv][rl];
                if (ok && bits == now && now < before) s_bestk[wv][rl] = 0x7fffffff;   // a new minimum this round
                wave_lds_sync();
                if (ok && bits == now) atomicMin(&s_bestk[wv][rl], node);
                wave_lds_sync();
            }
            npairs = 0;
        };
        for (int i = 0; i < kper; i += NU) {
            float d[NU];
            {
#pragma clang fp contract(fast)
#pragma unroll
                for (int u = 0; u < NU; u++) {
                    const int nd = node_lo + i + u < k ? node_lo + i + u : k - 1;
                    const float *wr = (WLDS ? s_w32 : w32) + (size_t)nd * (8 * CB);   // rows are zero-padded to 8 * CB channels
                    f32x2 acc[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};   // independent chains
#pragma unroll
                    for (int q = 0; q < 4 * CB; q++) {
                        const f32x2 wv2 = {wr[2 * q], wr[2 * q + 1]};
                        const f32x2 xx = {xv[2 * q], xv[2 * q + 1]};
                        const f32x2 t = xx - wv2;
                        acc[q & 3] = t * t + acc[q & 3];
                    }
                    const f32x2 sum = (acc[0] + acc[1]) + (acc[2] + acc[3]);
                    d[u] = sum.x + sum.y;
                }
            }
#pragma unroll
            for (int u = 0; u < NU; u++) {
                const int nd = node_lo + i + u;
                const bool flag = !skip && nd < node_hi && nd != ref && !(d[u] > thr);   // a NaN stays in: binary64 decides
                const unsigned long long m = __ballot(flag);
                if (m) {
                    const unsigned ahead = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                    if (flag) s_pairs[wv][npairs + ahead] = ((unsigned)slot << 16) | (unsigned)nd;
                    npairs += (unsigned)__popcll(m);
                    if (npairs > (unsigned)(kPairCap - 64)) {
                        wave_lds_sync();
                        flush();
                    }
                }
            }
        }
        wave_lds_sync();
        flush();
        if (valid && grp == 0) {
            const int win = s_bestk[wv][slot];
            labels[row] = win == 0x7fffffff ? 0 : win + 1;
        }
        wave_lds_sync();
    }
}

// ------------------------------------------------------------------------------------------------
// 3c. WIDE rows (c > 128: cell x pixel-cluster count tables over the 400 nodes of a 20 x 20 pixel SOM,
//     cell_cluster_utils.py:63-192 -> cluster_helpers.py:304-416).  No filter: every row is evaluated in the
//     oracle's arithmetic (binary64, j ascending, no contraction, sqrt, first strict minimum).  Cell tables hold
//     10^5 .. 10^6 rows, so the plain form is enough: a wave takes RB rows at a time, stages them in LDS (coalesced
//     loads; every later read of x_j is an LDS broadcast), lanes <-> nodes read consecutive nodes of channel j from
//     the transposed codebook [c][k] (workspace, L2-resident: 320 KB at 100 x 400) and share each value among the RB
//     rows.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wide_transpose_kernel(const double *__restrict__ w, int k, int c, double *__restrict__ wt,
                                                             AssignHdr *hdr, unsigned n)
{
    for (int e = blockIdx.x * 256 + threadIdx.x; e < k * c; e += gridDim.x * 256) {
        const int j = e / k, node = e - j * k;
        wt[e] = w[(size_t)node * c + j];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        hdr->amb_count = n;        // every row takes the exact arithmetic (pxsom_assign_last_exact_rows)
        hdr->force_exact = 1;
    }
}

constexpr int kWideRows = 4;   // rows a wave evaluates together
template <typename T>
__global__ __launch_bounds__(256) void bmu_wide_kernel(const T *__restrict__ x, int64_t n, int c, int64_t ldx,
                                                       const double *__restrict__ wt, int k, int32_t *__restrict__ labels)
{
    extern __shared__ double wide_rows[];                       // [4 waves][kWideRows][c]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double *mine = wide_rows + (size_t)wv * kWideRows * c;
    const int64_t nbatches = (n + kWideRows - 1) / kWideRows;
    for (int64_t b = (int64_t)blockIdx.x * 4 + wv; b < nbatches; b += (int64_t)gridDim.x * 4) {
        const int64_t row0 = b * kWideRows;
#pragma unroll
        for (int u = 0; u < kWideRows; u++) {
            const int64_t row = row0 + u < n ? row0 + u : n - 1;      // surplus slots redo the last row
            const T *rp = x + row * ldx;
            for (int j = lane; j < c; j += 64) mine[(size_t)u * c + j] = (double)rp[j];
        }
        wave_lds_sync();
        double best[kWideRows];
        int bestk[kWideRows];
#pragma unroll
        for (int u = 0; u < kWideRows; u++) {
            best[u] = DBL_MAX;
            bestk[u] = 0x7fffffff;
        }
        for (int base = 0; base < k; base += 128) {                  // two node slots per lane and sweep
            const int n0 = base + lane, n1 = base + 64 + lane;
            const int c0 = n0 < k ? n0 : k - 1, c1 = n1 < k ? n1 : k - 1;
            double d0[kWideRows], d1[kWideRows];
#pragma unroll
            for (int u = 0; u < kWideRows; u++) d0[u] = d1[u] = 0.0;
            for (int j0 = 0; j0 < c; j0 += 4) {                      // four channels' codebook values requested together
                double wa[4], wb[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int j = j0 + i < c ? j0 + i : c - 1;
                    wa[i] = wt[(size_t)j * k + c0];
                    wb[i] = wt[(size_t)j * k + c1];
                }
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    if (j0 + i < c) {   // uniform
#pragma unroll
                        for (int u = 0; u < kWideRows; u++) {
                            const double xj = mine[(size_t)u * c + j0 + i];
                            const double t0 = xj - wa[i], t1 = xj - wb[i];
                            d0[u] += t0 * t0;
                            d1[u] += t1 * t1;
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < kWideRows; u++) {                    // ascending node order within the lane
                const double s0 = sqrt(d0[u]), s1 = sqrt(d1[u]);
                if (n0 < k && s0 < best[u]) {
                    best[u] = s0;
                    bestk[u] = n0;
                }
                if (n1 < k && s1 < best[u]) {
                    best[u] = s1;
                    bestk[u] = n1;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < kWideRows; u++) {
            const double smin = pxsom::wave_min_f64(best[u]);
            const unsigned cand = best[u] == smin ? (unsigned)bestk[u] : 0xffffffffu;
            const int win = (int)pxsom::wave_min_u32(cand);          // 0x7fffffff: no finite distance (NaN row)
            if (lane == 0 && row0 + u < n) labels[row0 + u] = win == 0x7fffffff ? 0 : win + 1;
        }
        wave_lds_sync();                                             // the rows are overwritten by the next batch
    }
}

// distance of every row to its labelled node (only when the caller asks for dists)
template <typename T>
__global__ __launch_bounds__(256) void bmu_dist_kernel(const T *__restrict__ x, int64_t n, int c,
                                                       int64_t ldx, const double *__restrict__ w,
                                                       const int32_t *__restrict__ labels,
                                                       double *__restrict__ dist)
{
    for (int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x; row < n;
         row += (int64_t)gridDim.x * 256) {
        const int lab = labels[row];
        double d = DBL_MAX;
        if (lab > 0) {
            const T *rp = x + row * ldx;
            const double *wp = w + (size_t)(lab - 1) * c;
            double xdist = 0.0;
            for (int j = 0; j < c; j++) {
                const double tmp = (double)rp[j] - wp[j];
                xdist += tmp * tmp;
            }
            d = sqrt(xdist);
        }
        dist[row] = d;
    }
}
#pragma clang fp contract(fast)

// The prep kernel (one workgroup) stages the codebook in LDS whenever it fits beside its own tables: from
// global memory its dependent reads cost the L2 latency each (55 us at K = 400, C = 40).
// 1024 threads once the codebook has more than 128 nodes or 4096 entries
static bool prep_wide(int k, int c) { return k > 128 || k * c > 4096; }

static int prep_stage(size_t stage_bytes)
{
    constexpr size_t kPrepStageMax = 132 * 1024;
    static pxsom::PerDevice<bool> raised_on;
    bool &raised = raised_on.here();
    if (!raised) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(bmu_prep_kernel<256>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPrepStageMax);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(bmu_prep_kernel<1024>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPrepStageMax);
        raised = true;
    }
    return stage_bytes <= kPrepStageMax;
}

template <typename T>
int assign_typed(const T *x, int64_t n, int c, int64_t ldx, const double *w, int k, int32_t *labels,
                 double *dist, char *ws, const Layout &L_in, hipStream_t st, double *stats = nullptr,
                 bool prepared = false, FinishTables *fin = nullptr, bool screen_all = false)
{
    // binary16 rows of a wide codebook: packed-K fragments (pxsom_assign.h packed_k).  `prepared`: the caller's layout
    // says what the workspace holds.
    const int npk = prepared ? L_in.npk : (packed_rows_ok<T>(x, ldx) && !stats ? packed_k(c, k, sizeof(T) == 2) : 0);
    const Layout L = make_layout(n, c, k, npk);
    // (a scheduled step's rows where they lie, pxsom::RowView: the streamed filters and the exact kernels take views; the
    // register-resident kernels, the all-exact route of rows wider than 128 and the distance output do not)
    if (pxsom::row_view_active() && (c > kFilterMaxChannels || L.nch == 1 || dist))
        return pxsom::fail(PXSOM_ERR_UNSUPPORTED, "assign: a row view on a shape whose kernels do not take views");
    if (c > kFilterMaxChannels) {   // wide rows: no filter, every row in the oracle's arithmetic (section 3c)
        double *wt = reinterpret_cast<double *>(ws + L.off_wt);
        hipLaunchKernelGGL(wide_transpose_kernel, dim3((unsigned)std::min<int64_t>(((int64_t)k * c + 255) / 256, 1024)), dim3(256), 0, st,
                           w, k, c, wt, reinterpret_cast<AssignHdr *>(ws), (unsigned)n);
        PXSOM_LAUNCH_CHECK("wide_transpose_kernel");
        const size_t lds = (size_t)4 * kWideRows * c * sizeof(double);
        auto kern = bmu_wide_kernel<T>;
        static pxsom::PerDevice<size_t> raised;
        size_t &have = raised.here();
        if (lds > 48 * 1024 && have < lds) {
            PXSOM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            have = lds;
        }
        const int wcus = pxsom::device_cu_count();
        const int64_t wgrid = std::max<int64_t>(1, std::min<int64_t>((n + 4 * kWideRows - 1) / (4 * kWideRows), (int64_t)wcus * 4));
        pxsom::Prof *wprof = pxsom::current_prof();
        pxsom::prof_mark(wprof, st, true, n);
        PXSOM_TIMED_LAUNCH(kern, dim3((unsigned)wgrid), dim3(256), lds, st, x, n, c, ldx, wt, k, labels);
        pxsom::prof_mark(wprof, st, false, n);
        PXSOM_LAUNCH_CHECK("bmu_wide_kernel");
        if (dist) {
            int dgrid = (int)std::min<int64_t>((n + 255) / 256, (int64_t)wcus * 8);
            hipLaunchKernelGGL(bmu_dist_kernel<T>, dim3(dgrid < 1 ? 1 : dgrid), dim3(256), 0, st, x, n, c, ldx, w, labels, dist);
            PXSOM_LAUNCH_CHECK("bmu_dist_kernel");
        }
        return PXSOM_OK;
    }
    if constexpr (sizeof(T) == 8) {
        // binary64 rows of the register-resident shapes (what the drop-in classes label): one self-contained launch of the
        // two-tile kernel -- prepares the codebook itself and settles its listed rows inside (round 5: no spills, unlike
        // bmu_filter_fast<double>); the workspace only records that no row was listed
        if (!stats && onepass_labels_route(sizeof(T)) && filter_fast_path<T>(x, n, c, ldx, L)) {   // (prepared or not: it prepares itself)
            PXSOM_HIP_TRY(hipMemsetAsync(ws, 0, sizeof(unsigned), st));
            pxsom::Prof *prof1 = pxsom::current_prof();
            pxsom::prof_mark(prof1, st, true, n);
            launch_onepass_labels(x, n, c, ldx, L, labels, w, st);
            pxsom::prof_mark(prof1, st, false, n);
            PXSOM_LAUNCH_CHECK("bmu_onepass_kernel");
            if (dist) {
                int dgrid = (int)std::min<int64_t>((n + 255) / 256, (int64_t)pxsom::device_cu_count() * 8);
                if (dgrid < 1) dgrid = 1;
                hipLaunchKernelGGL(bmu_dist_kernel<T>, dim3(dgrid), dim3(256), 0, st, x, n, c, ldx, w, labels, dist);
                PXSOM_LAUNCH_CHECK("bmu_dist_kernel");
            }
            return PXSOM_OK;
        }
    }
    if (!prepared && !stats) {  // prepared: pxsom_batch_update_prepare did this; stats: the accumulating filter
                                 // prepares the codebook inside its own launch
        const size_t stage_bytes = (size_t)k * c * sizeof(double);
        const int stage = prep_stage(stage_bytes);
        hipLaunchKernelGGL(prep_wide(k, c) ? bmu_prep_kernel<1024> : bmu_prep_kernel<256>, dim3(1), dim3(prep_wide(k, c) ? 1024 : 256),
                           stage ? stage_bytes : 0, st, w, k, c,
                           reinterpret_cast<AssignHdr *>(ws),
                           reinterpret_cast<half8 *>(ws + L.off_wfrag), reinterpret_cast<f32x4 *>(ws + L.off_bias),
                           L.nb, L.nch, L.cpl, L.idx_bits, L.node_bits, stage, (double *)nullptr, 0,
                           L.has_wt() ? reinterpret_cast<double *>(ws + L.off_wt) : nullptr,
                           reinterpret_cast<float *>(ws + L.off_w32), L.cp32, L.npk,
                           // the filters centre rows and codebook (AssignHdr::mu_s) -- all but the streamed one on binary16
                           // rows, whose two-term split needs x * scale to BE a binary16 number
                           (filter_fast_path<T>(x, n, c, ldx, L) || (sizeof(T) != 2 && L.npk == 0)) ? 1 : 0);
        PXSOM_LAUNCH_CHECK("bmu_prep_kernel");
    }

    const int cus = pxsom::device_cu_count();
    pxsom::Prof *prof = pxsom::current_prof();
    pxsom::prof_mark(prof, st, true, n);
    if constexpr (sizeof(T) == 2) {
        if (L.npk > 0) launch_filter_packed(x, n, c, ldx, ws, L, labels, st, !prepared);
        else launch_filter_any<T>(x, n, c, ldx, ws, L, labels, stats, w, st, fin);
    } else {
        launch_filter_any<T>(x, n, c, ldx, ws, L, labels, stats, w, st, fin);
    }
    pxsom::prof_mark(prof, st, false, n);
    PXSOM_LAUNCH_CHECK("bmu_filter_kernel");

    if (stats) return PXSOM_OK;   // the accumulating filter settled its listed rows itself
    const size_t wt_bytes = (size_t)k * c * sizeof(double);
    const int use_lds = wt_bytes <= 64 * 1024;
    // listed rows are a small fraction of n and the kernel grid-strides over the list, but a mini-batch of a
    // few thousand rows can list hundreds in early training steps: enough workgroups that those still take one
    // or two rounds (workgroups without rows leave at once)
    int egrid = (int)std::min<int64_t>((n + 31) / 32, (int64_t)cus * 4);
    if (egrid < 1) egrid = 1;
    // the binary32 codebook copy: scalar cache while it fits it, LDS up to 64 KB (two workgroups per CU), L2 beyond
    const size_t w32_bytes = (size_t)k * L.cp32 * sizeof(float);
    const bool w32_lds = w32_bytes > 12 * 1024 && w32_bytes <= 64 * 1024;
    const unsigned screened_from = screen_min_rows(c, w32_lds, screen_all);
    hipLaunchKernelGGL(bmu_exact_kernel<T>, dim3(egrid), dim3(256), use_lds ? wt_bytes : 0, st, x, c, ldx, w,
                       k, reinterpret_cast<const AssignHdr *>(ws),
                       reinterpret_cast<const unsigned *>(ws + L.off_list), labels, use_lds,
                       reinterpret_cast<const double *>(ws + L.off_wt), screened_from, pxsom::current_row_view());
    PXSOM_LAUNCH_CHECK("bmu_exact_kernel");
    if ((uint64_t)n >= screened_from) {   // (an input shorter than the crossover cannot list that many rows)
        void (*kern)(const T *, int, int64_t, const double *, const float *, int, const AssignHdr *, const unsigned *, int32_t *,
                     unsigned, pxsom::RowView) = nullptr;
        switch (L.cp32 / 8) {
#define PXSOM_SCREENED(CB)                                                                                  \
    case CB:                                                                                                \
        kern = w32_lds ? bmu_exact_screened_kernel<T, CB, (CB <= 3 ? 4 : CB <= 6 ? 2 : 1), true>            \
                       : bmu_exact_screened_kernel<T, CB, (CB <= 3 ? 4 : CB <= 6 ? 2 : 1), false>;          \
        break;
            PXSOM_SCREENED(2) PXSOM_SCREENED(3) PXSOM_SCREENED(4) PXSOM_SCREENED(5) PXSOM_SCREENED(6) PXSOM_SCREENED(8)
            PXSOM_SCREENED(10) PXSOM_SCREENED(13) PXSOM_SCREENED(16)
#undef PXSOM_SCREENED
        }
        // every workgroup resident at once (the kernel sizes its batches by the waves of the launch)
        const int cb = L.cp32 / 8, by_regs = (cb <= 3 || (sizeof(T) <= 4 && cb <= 5)) ? 4 : 2;
        const int by_lds = w32_lds ? (int)std::max<size_t>(1, (160 * 1024) / (w32_bytes + 12 * 1024)) : by_regs;
        hipLaunchKernelGGL(kern, dim3(cus * std::min(by_regs, by_lds)), dim3(256), w32_lds ? w32_bytes : 0, st, x, c, ldx, w, reinterpret_cast<const float *>(ws + L.off_w32), k,
                           reinterpret_cast<const AssignHdr *>(ws), reinterpret_cast<const unsigned *>(ws + L.off_list), labels,
                           screened_from, pxsom::current_row_view());
        PXSOM_LAUNCH_CHECK("bmu_exact_screened_kernel");
    }

    if (dist) {
        int dgrid = (int)std::min<int64_t>((n + 255) / 256, (int64_t)cus * 8);
        if (dgrid < 1) dgrid = 1;
        hipLaunchKernelGGL(bmu_dist_kernel<T>, dim3(dgrid), dim3(256), 0, st, x, n, c, ldx, w, labels, dist);
        PXSOM_LAUNCH_CHECK("bmu_dist_kernel");
    }
    return PXSOM_OK;
}

}  // namespace

PXSOM_EXPORT size_t pxsom_assign_workspace_bytes(int64_t n, int c, int k)
{
    if (n < 0 || c < 1 || c > PXSOM_MAX_CHANNELS || k < 1 || k > PXSOM_MAX_NODES) return 0;
    return make_layout(n, c, k).total;
}

PXSOM_EXPORT int pxsom_assign(const void *x_dev, int64_t n, int c, int64_t ldx, int dtype,
                              const double *w_dev, int k, int32_t *labels_dev, double *dist_dev,
                              void *workspace_dev, size_t workspace_bytes, void *stream)
{
    return pxsom_assign_ex(x_dev, n, c, ldx, dtype, w_dev, k, labels_dev, dist_dev, workspace_dev, workspace_bytes, 0, stream);
}

PXSOM_EXPORT int pxsom_assign_ex(const void *x_dev, int64_t n, int c, int64_t ldx, int dtype,
                                 const double *w_dev, int k, int32_t *labels_dev, double *dist_dev,
                                 void *workspace_dev, size_t workspace_bytes, int flags, void *stream)
{
    if (n < 0 || n > 0x7fffffffLL)
        return pxsom::fail(PXSOM_ERR_INVALID_ARG, "pxsom_assign: n=%lld outside [0, 2^31)", (long long)n);
    if (c < 1 || c > PXSOM_MAX_CHANNELS)
        return pxsom::fail(PXSOM_ERR_UNSUPPORTED, "pxsom_assign: c=%d outside [1, %d]", c, PXSOM_MAX_CHANNELS);
    if (k < 1 || k > PXSOM_MAX_NODES)
        return pxsom::fail(PXSOM_ERR_UNSUPPORTED, "pxsom_assign: k=%d outside [1, %d]", k, PXSOM_MAX_NODES);
    if (ldx < c) return pxsom::fail(PXSOM_ERR_INVALID_ARG, "pxsom_assign: ldx=%lld < c=%d", (long long)ldx, c);
    if (!pxsom::dtype_ok(dtype)) return pxsom::fail(PXSOM_ERR_UNSUPPORTED, "pxsom_assign: dtype %d", dtype);
    if (!w_dev || (n > 0 && (!x_dev || !labels_dev)))
        return pxsom::fail(PXSOM_ERR_INVALID_ARG, "pxsom_assign: null pointer");
    const Layout L = make_layout(n, c, k);
    if (!workspace_dev || workspace_bytes < L.total)
        return pxsom::fail(PXSOM_ERR_WORKSPACE, "pxsom_assign: workspace %zu < %zu bytes", workspace_bytes,
                           L.total);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    char *ws = reinterpret_cast<char *>(workspace_dev);
    if (n == 0) {
        PXSOM_HIP_TRY(hipMemsetAsync(ws, 0, kHdrBytes, st));
        return PXSOM_OK;
    }
    const bool screen_all = (flags & PXSOM_ASSIGN_SCREEN_ALL_LISTS) != 0;
    PXSOM_DISPATCH_DTYPE(dtype, x_dev, xp,
                         assign_typed<T>(xp, n, c, ldx, w_dev, k, labels_dev, dist_dev, ws, L, st, nullptr, false, nullptr, screen_all));
}

int pxsom_bmu::assign_accumulate(const void *x_dev, int64_t n, int c, int64_t ldx, int dtype, const double *w_dev,
                                 int k, int32_t *labels_dev, double *stats_dev, void *workspace_dev,
                                 size_t workspace_bytes, hipStream_t st, bool *fused, FinishTables *fin)
{
    *fused = false;
    if (n < 64 || n > 0x7fffffffLL || c < 1 || c > PXSOM_MAX_CHANNELS || k < 1 || k > PXSOM_MAX_NODES || ldx < c ||
        !x_dev || !w_dev || !labels_dev || !stats_dev || !workspace_dev)
        return PXSOM_OK;  // the unfused route reports what is wrong with the arguments
    const Layout L = make_layout(n, c, k);
    if (workspace_bytes < L.total) return PXSOM_OK;
    char *ws = reinterpret_cast<char *>(workspace_dev);
    if (!pxsom::dtype_ok(dtype)) return PXSOM_OK;
    PXSOM_DISPATCH_DTYPE(dtype, x_dev, xp,
                         (filter_fast_path<T>(xp, n, c, ldx, L)
                              ? (*fused = true, assign_typed<T>(xp, n, c, ldx, w_dev, k, labels_dev, nullptr, ws, L, st,
                                                                stats_dev, false, fin))
                              : PXSOM_OK));
}

// pxsom_assign on a workspace that pxsom_batch_update_prepare prepared for w_dev (no prep launch)
int pxsom_bmu::assign_prepared(const void *x_dev, int64_t n, int c, int64_t ldx, int dtype, const double *w_dev,
                               int k, int32_t *labels_dev, void *workspace_dev, size_t workspace_bytes,
                               hipStream_t st, int npk)
{
    const Layout L = make_layout(n, c, k, npk);
    if (!workspace_dev || workspace_bytes < L.total)
        return pxsom::fail(PXSOM_ERR_WORKSPACE, "pxsom_batch_accumulate: workspace %zu < %zu bytes", workspace_bytes,
                           L.total);
    char *ws = reinterpret_cast<char *>(workspace_dev);
    PXSOM_DISPATCH_DTYPE(dtype, x_dev, xp,
                         assign_typed<T>(xp, n, c, ldx, w_dev, k, labels_dev, nullptr, ws, L, st, nullptr, true));
}

// prep alone, optionally clearing the batch statistics (pxsom_batch_update_prepare)
int pxsom_bmu::prepare_only(const double *w_dev, int c, int k, void *workspace_dev, size_t workspace_bytes,
                            double *zero_stats, hipStream_t st)
{
    const Layout L = make_layout(0, c, k);
    if (workspace_bytes < L.total) return pxsom::fail(PXSOM_ERR_WORKSPACE, "prepare: workspace too small");
    char *ws = reinterpret_cast<char *>(workspace_dev);
    const size_t stage_bytes = (size_t)k * c * sizeof(double);
    const int stage = prep_stage(stage_bytes);
    hipLaunchKernelGGL(prep_wide(k, c) ? bmu_prep_kernel<1024> : bmu_prep_kernel<256>, dim3(1), dim3(prep_wide(k, c) ? 1024 : 256),
                       stage ? stage_bytes : 0, st, w_dev, k, c,
                       reinterpret_cast<AssignHdr *>(ws), reinterpret_cast<half8 *>(ws + L.off_wfrag),
                       reinterpret_cast<f32x4 *>(ws + L.off_bias), L.nb, L.nch, L.cpl, L.idx_bits, L.node_bits, stage,
                       zero_stats, zero_stats ? k * (c + 1) : 0,
                       L.has_wt() ? reinterpret_cast<double *>(ws + L.off_wt) : nullptr,
                           reinterpret_cast<float *>(ws + L.off_w32), L.cp32, 0, 0);
    PXSOM_LAUNCH_CHECK("bmu_prep_kernel");
    return PXSOM_OK;
}

PXSOM_EXPORT int pxsom_assign_last_exact_rows(const void *workspace_dev, void *stream, int64_t *out_rows)
{
    if (!workspace_dev || !out_rows) return pxsom::fail(PXSOM_ERR_INVALID_ARG, "last_exact_rows: null");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    unsigned v = 0;
    PXSOM_HIP_TRY(hipMemcpyAsync(&v, workspace_dev, sizeof(unsigned), hipMemcpyDeviceToHost, st));
    PXSOM_HIP_TRY(hipStreamSynchronize(st));
    *out_rows = (int64_t)v;
    return PXSOM_OK;
}
