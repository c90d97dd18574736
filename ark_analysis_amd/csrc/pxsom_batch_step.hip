// pxsom_batch_step.hip -- the fused mini-batch step of the batch SOM rule (K6b), one launch per step.
//
// Batch rule (no pyFlowSOM analogue; oracle of record oracle/pxsom_oracle.c orc_som_batch; the reference call it
// stands in for is PixieSOMCluster.train_som, /root/reference/src/ark/phenotyping/cluster_helpers.py:98-116):
//     step g:  W_g = update(W_{g-1}, stats_{g-1});  b_i = BMU(x_i, W_g);  stats_g[b_i] += [x_i, 1]
// A step is latency-bound (16 K rows x 88 B against ~2 us of HBM latency), so the kernel is built around the
// length of its dependency chain, not around bandwidth:
//   * every workgroup (512 threads) applies the pending update and prepares the codebook for itself -- no update
//     launch, no prep launch, no second pass over a workspace.  Update: separable window sums in registers
//     (orc_batch_update's order, bit for bit), new node values stay in the registers of the thread that formed
//     them: thread <-> (node, lane group q) holds exactly the channels of one MFMA A-fragment element, so norms,
//     the duplicate key, the fp16 hi/lo split and the fragment store need no further LDS round trip;
//   * the rows of the step are requested before any of that (they do not depend on the codebook);
//   * BMU search: fp16-split MFMA filter (the K7 scheme: v_mfma_f32_16x16x32_f16, nodes on the M axis, running
//     top-2 with the node id in the low mantissa bits, rigorous tolerance) on 16-row tiles, TPW tiles per wave;
//     rows it is sure of go straight into the workgroup's binary64 table in LDS (ds_add_f64); rows it is not sure
//     of are queued IN LDS with their values and settled afterwards in the oracle's own arithmetic (binary64,
//     j ascending, no contraction, sqrt, first strict minimum) by whichever wave is free;
//   * one staggered flush of the table with device-scope atomics; the statistics buffer of the NEXT step is
//     cleared here (three buffers rotate), W_g is written to the twin codebook buffer by workgroup 0.
// Shapes: 10 x 10 grid, even c <= 32, rows 2-element aligned (the Pixie pixel SOM; BASELINE.json configs 2, 3).
#include <algorithm>
#include <cfloat>
#include <cmath>

#include "pxsom_batch_step.h"

namespace pxsom_bmu {
namespace {

#ifdef PXSOM_PHASE_TIMING
__device__ long long g_block_ticks[2 * 256];   // per workgroup: first and last instruction (s_memtime)
#endif

constexpr int kStepThreads = 512, kStepWaves = 8;
// Copies of the workgroup's statistics table (pixel lane pix adds into copy pix % copies) against same-word contention of
// the ds_add_f64 lanes while the codebook is crowded.  Measured in round 3 with 4 copies: the crowded head steps did not
// move (36.0 / 33.1 us before and after: the LDS serialisation is not what holds them) and every step paid for clearing and
// adding up the copies (tail step 11.4 -> 12.1 us): one table.
__host__ __device__ inline int table_copies(int c) { return (void)c, 1; }
// Row stride of the table in 8-byte words: the channel count padded to an ODD number.  The 64 lanes of one ds_add_f64 hit words
// node * stride + (lane group) * CPL + 2 p for 16 unrelated nodes: with stride 22 the nodes spread over 8 of the 16 double-word
// bank pairs (gcd(22, 16) = 2), with 23 over all of them (the one-pass kernel's acc_stride: same reason).
__host__ __device__ inline int table_stride(int c) { return c | 1; }

// LDS carve-up (bytes from the start of the dynamic segment)
struct StepLds {
    size_t ls, wt, tl, key, red, ovf, frag, bias, hdr, mu, total;
};
__host__ __device__ inline StepLds step_lds(int c)
{
    StepLds L;
    size_t o = 0;
    L.ls = o;    o += ((size_t)kK * table_stride(c) + kK) * 8 * table_copies(c);   // copies x table [K x stride sums | K counts]
    L.wt = o;    o += (size_t)c * kK * 8;                 // codebook, transposed [c][K]
    L.tl = o;    o += (size_t)kK * (c + 1) * 8;           // window-sum scratch; later the queue of listed rows
    L.key = o;   o += (size_t)kK * 8;
    L.red = o;   o += 2 * kStepWaves * 8;
    L.ovf = o;   o += (size_t)kStepWaves * 32 * 8;        // one row per wave (queue overflow)
    L.frag = o;  o += (size_t)kNB * 2 * 64 * 16;          // [NB][hi, lo][64] half8
    L.bias = o;  o += (size_t)kNB * 64 * 16;              // [NB][64] f32x4
    L.hdr = o;   o += 64;
    L.mu = o;    o += 40 * 4;                             // the run's centring vector (32 words) and its norm (word 32)
    L.total = o;
    return L;
}

// BMU: the launch is a BMU-only step (pending update with its threshold pinned at 0.5) and the windowed update is compiled out;
// !BMU: the other way round.  The same instructions run on a step whichever kernel it takes, but a kernel that CONTAINS both
// updates takes 10.6 us per BMU-only step where the specialised one takes 9.3, and 0.2 - 1.4 us more per windowed step
// (profiles/r04/step_bmu_only_specialisation.txt).
// EXCH (round 5): the instantiation for ranks of a sharded run on a peer-to-peer communicator -- the rule's exchange runs inside
// the launch (StepArgs::xch_*): no all-reduce launch between two steps.
#ifdef PXSOM_STEP_COUNT_LISTED   // (counting build only, scripts/dev/listed_per_step.py: rows each launch lists for the exact path)
__device__ unsigned g_dbg_listed[4096], g_dbg_listed_max[4096], g_dbg_rows[4096], g_dbg_launch;
__global__ void dbg_next_launch() { g_dbg_launch = g_dbg_launch + 1; }
#endif
template <typename T, int CPL, int TPW, bool BMU, bool EXCH = false>
__global__ __launch_bounds__(kStepThreads) void batch_step_kernel(const T *__restrict__ x, int64_t n, int c, int64_t ldx,
                                                                  double *__restrict__ stats, StepArgs sa)
{
    extern __shared__ __attribute__((aligned(16))) char step_smem[];
    const StepLds L = step_lds(c);
    double *ls = reinterpret_cast<double *>(step_smem + L.ls);
    double *wt = reinterpret_cast<double *>(step_smem + L.wt);
    double *tl = reinterpret_cast<double *>(step_smem + L.tl);
    unsigned long long *key = reinterpret_cast<unsigned long long *>(step_smem + L.key);
    double *red = reinterpret_cast<double *>(step_smem + L.red);
    double *ovf = reinterpret_cast<double *>(step_smem + L.ovf);
    half8 *frag_l = reinterpret_cast<half8 *>(step_smem + L.frag);
    f32x4 *bias_l = reinterpret_cast<f32x4 *>(step_smem + L.bias);
    StepHdr *hdr = reinterpret_cast<StepHdr *>(step_smem + L.hdr);
    float *mu_l = reinterpret_cast<float *>(step_smem + L.mu);

    constexpr int NP = CPL / 2;
    typedef typename Pair<T>::type P2;
    PXSOM_PHASE(8);
#ifdef PXSOM_PHASE_TIMING
    const long long t_start = wall_clock64();   // s_memrealtime: 100 MHz, one counter for the whole chip
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pix = lane & 15, q = lane >> 4;
    constexpr int kRowsPerWave = 16 * TPW, kRowsPerWg = kStepWaves * kRowsPerWave;

    // ---- P0: everything that does not depend on the codebook is requested first --------------------------
    // rows of this wave's tiles (rows past the end re-read the last row and are ignored afterwards)
    P2 raw[TPW][NP];
    const int cs = table_stride(c);
    const size_t tstride = (size_t)kK * cs + kK;     // doubles per copy of the statistics table
    const int ncopies = table_copies(c);
    const unsigned group_w = (unsigned)sa.group_w;
    const double qmagic = sizeof(T) == 8 ? sa.qmagic : 0.0;
    auto load_rows = [&](int64_t blk) {
#pragma unroll
        for (int t = 0; t < TPW; t++) {
            int64_t row = blk * kRowsPerWg + (int64_t)wv * kRowsPerWave + t * 16 + pix;
            if (row > n - 1) row = n - 1;
            const T *rp;
            if (group_w == 1u) {
                rp = x + row * ldx;
            } else {   // a scheduled step: group_w consecutive rows out of every `phases` (rows per step < 2^32: host check)
                const unsigned grp = __umulhi((unsigned)row, sa.group_magic) >> sa.group_shift, sub = (unsigned)row - grp * group_w;
                rp = x + (int64_t)grp * sa.group_stride + (int64_t)sub * ldx;
            }
#pragma unroll
            for (int p = 0; p < NP; p++) {
                int ch = q * CPL + 2 * p;
                if (ch > c - 2) ch = c - 2;   // slots past c re-read the last valid pair: their codebook slots are zero
                if constexpr (sizeof(T) == 2) {
                    const half2_t h = *reinterpret_cast<const half2_t *>(rp + ch);
                    raw[t][p].x = h[0];
                    raw[t][p].y = h[1];
                } else {
                    raw[t][p] = *reinterpret_cast<const P2 *>(rp + ch);
                }
            }
        }
    };
    const int64_t nblocks = (n + kRowsPerWg - 1) / kRowsPerWg;
    int64_t blk = blockIdx.x;

    // thread <-> (node, lane group nq): the channels nq*CPL .. nq*CPL + CPL - 1 of one node
    const int node = tid >> 2, nq = tid & 3;
    const bool has_node = node < kK;
    double wv_[CPL];   // this thread's node values: old, then new
    // the run's centring vector (33 words of HBM): one word per thread of the first wave, requested with everything else,
    // parked in LDS before the first barrier; prep and search read it from there
    float mu_word = 0.f;
    const int NC = c + 1;
    auto load_centring = [&]() {
        if (tid < 33 && sa.mu32) mu_word = sa.mu32[tid < 32 ? tid : kFilterMaxChannels];   // 32 channels + the vector's norm
    };
    auto park_centring = [&]() {
        if (tid < 40) mu_l[tid] = mu_word;
    };
    // Order of the requests matters: vmcnt retires loads in issue order, so what is needed first is asked for
    // first -- the statistics (pass 1), then the old node values (P3), the step's rows (HBM, slowest) last.
    // BMU-only steps (the tail of a pass: thr = 0.5, r = 0): the window of a node is the node itself -- both separable passes
    // would add +-0 to S[x][y], the same bits -- so thread <-> (node, lane group) asks for ITS words of the statistics directly
    // (no scratch pass, no barrier before them), and the gain 1 - (1 - alpha)^n with its chain of products and the reciprocal are formed
    // once per node by the first two waves (lane <-> node) instead of by all seven waves that hold node lanes: the update of
    // a tail step was binary64 issue on two waves per SIMD (profiles/r03/step_phase_timing.txt: 1.6 us of 9.1).
    // word e of the pending update's statistics: this rank's buffer, or -- EXCH with a wait epoch -- the sum of the ranks' slots
    // in rank order, starting from +0.0 (the same additions p2p_allreduce_kernel makes: the same bits on every rank)
    char *xch_mine = nullptr;
    const int xch_parity = (int)(sa.xch_wait & 1ull);
    // Round 6 (advisor, round 5): a workgroup that meets a late peer does NOT leave the launch -- it records the epoch in the
    // block's error word (first one kept; ANY workgroup, not only the first), searches no rows, poisons the codebook it would
    // hand on, and still takes its ticket at the end, marked late: the last workgroup through the ticket then resets it and
    // raises this rank's flag of the next epoch over slots full of NaN -- the peers fail fast and loud instead of waiting four
    // seconds in every later step of the call, and no ticket count survives into the next launch.
    bool xch_late = false;
    if constexpr (EXCH) {
        if (sa.xch_wait) {
            xch_mine = sa.xch_peers[sa.xch_rank];
            __shared__ int s_xch_late;
            // (an epoch already on record: a peer was late earlier in this call -- every later step would wait its four seconds
            // for the same peer; it is late at once instead)
            if (tid == 0)
                s_xch_late = __hip_atomic_load(&reinterpret_cast<pxsom::P2PBlock *>(xch_mine)->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0ull;
            __syncthreads();
            if (tid < sa.xch_nranks && !s_xch_late) {
                pxsom::P2PBlock *blk_mine = reinterpret_cast<pxsom::P2PBlock *>(xch_mine);
                const long long t0 = (long long)wall_clock64();   // 100 MHz
                while (__hip_atomic_load(&blk_mine->flags[xch_parity][tid], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != sa.xch_wait) {
                    __builtin_amdgcn_s_sleep(2);
                    if ((long long)wall_clock64() - t0 > 400000000ll) {   // 4 s, as the separate exchange
                        s_xch_late = 1;
                        break;
                    }
                }
            }
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
            if (s_xch_late) {   // a peer never arrived: nothing to apply -- record the epoch (first one kept)
                xch_late = true;
                if (tid == 0) {
                    unsigned long long none = 0ull;
                    (void)__hip_atomic_compare_exchange_strong(&reinterpret_cast<pxsom::P2PBlock *>(xch_mine)->error, &none, sa.xch_wait,
                                                               __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
        }
    }
    auto stat = [&](size_t e) -> double {
        if constexpr (EXCH) {
            if (xch_mine) {
                double acc = 0.0;
                for (int p = 0; p < sa.xch_nranks; p++)
                    acc += __hip_atomic_load(pxsom::p2p_slot(xch_mine, xch_parity, p, sa.xch_nranks, (size_t)sa.xch_max_count) + e,
                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                return acc;
            }
        }
        return sa.stats_prev[e];
    };
    const int upd_r = sa.thr < 0.0 ? -1 : (sa.thr > 1.0e6 ? 1000000 : (int)floor(sa.thr));
    constexpr bool bmu_only = BMU;   // (the launch picks the kernel by the pending update's threshold: launch_step)
    double sdir[CPL];   // bmu_only: this thread's words of the statistics
    if (bmu_only) {
        double cnt = 0.0;
        if (tid < 2 * 64) cnt = stat((size_t)kK * c + (tid < kK ? tid : 0));
#pragma unroll
        for (int i = 0; i < CPL; i++) {
            const int ch = nq * CPL + i;
            sdir[i] = stat((has_node && ch < c) ? (size_t)node * c + ch : 0);
        }
#pragma unroll
        for (int i = 0; i < CPL; i++) {
            const int ch = nq * CPL + i;
            wv_[i] = sa.w_in[(has_node && ch < c) ? (size_t)node * c + ch : 0];
        }
        load_centring();
        if (blk < nblocks) load_rows(blk);
        PXSOM_PHASE(9);
        for (int e = tid; e < (kK * cs + kK) * ncopies; e += kStepThreads) ls[e] = 0.0;
        if (tid == 0) {
            hdr->q_n = 0u;
            hdr->bad = 0;
        }
        if (tid < 64) bias_l[6 * 64 + tid] = f32x4{kNegBig, kNegBig, kNegBig, kNegBig};   // rows of the last block without a node
        if (sa.stats_zero) {
            const int per = (sa.zero_count + (int)gridDim.x - 1) / (int)gridDim.x;
            const int z1 = min(((int)blockIdx.x + 1) * per, sa.zero_count);
            for (int e = (int)blockIdx.x * per + tid; e < z1; e += kStepThreads) sa.stats_zero[e] = 0.0;
        }
        park_centring();
        if (tid < kK) {   // tl[node] = gain (or -1: no rows, the node stays), tl[K + node] = 1 / n
#pragma clang fp contract(off)
            const double den = 0.0 + cnt;
            tl[tid] = den > 0.0 ? batch_gain(den, sa.q, sa.sat) : -1.0;
            tl[kK + tid] = den > 0.0 ? 1.0 / den : 0.0;
        }
        PXSOM_PHASE(10);
        PXSOM_PHASE(11);
        __syncthreads();
    } else if (sa.has_update) {
        // pass 1 of the separable window sums: thread <-> (grid row gx, column cc); column c carries the counts
        double S[kYD];
        const bool p1 = tid < kXD * NC;
        const int gx = p1 ? tid / NC : 0, cc = p1 ? tid - gx * NC : 0;
#pragma unroll
        for (int y = 0; y < kYD; y++)   // branch-free: idle threads re-read a valid word
            S[y] = stat(cc < c ? (size_t)(gx * kYD + y) * c + cc : (size_t)kK * c + gx * kYD + y);
#pragma unroll
        for (int i = 0; i < CPL; i++) {
            const int ch = nq * CPL + i;
            wv_[i] = sa.w_in[(has_node && ch < c) ? (size_t)node * c + ch : 0];
        }
        load_centring();
        if (blk < nblocks) load_rows(blk);
        PXSOM_PHASE(9);
        // (while those are in flight) clear the table and this workgroup's slice of the next buffer
        for (int e = tid; e < (kK * cs + kK) * ncopies; e += kStepThreads) ls[e] = 0.0;
        if (tid == 0) {
            hdr->q_n = 0u;
            hdr->bad = 0;
        }
        if (tid < 64) bias_l[6 * 64 + tid] = f32x4{kNegBig, kNegBig, kNegBig, kNegBig};   // rows of the last block without a node
        if (sa.stats_zero) {
            const int per = (sa.zero_count + (int)gridDim.x - 1) / (int)gridDim.x;
            const int z1 = min(((int)blockIdx.x + 1) * per, sa.zero_count);
            for (int e = (int)blockIdx.x * per + tid; e < z1; e += kStepThreads) sa.stats_zero[e] = 0.0;
        }
        park_centring();
        const double thr = sa.thr;
        const int r = thr < 0.0 ? -1 : (thr > 1.0e6 ? 1000000 : (int)floor(thr));
        double md[kXD > kYD ? kXD : kYD];
#pragma unroll
        for (int d = 0; d < (kXD > kYD ? kXD : kYD); d++) md[d] = d <= r ? 1.0 : 0.0;
        // BMU-only steps (the tail of a pass: thr = 0.5, r = 0): the window of a node is the node, both passes would add
        // +-0 to S[x][y] -- the same bits -- so the statistics go to the scratch as they are (one barrier instead of two,
        // no chains of 10 + 10 dependent additions)
        if (r == 0) {
            if (p1) {
#pragma unroll
                for (int y = 0; y < kYD; y++) tl[(size_t)(y * kXD + gx) * NC + cc] = 0.0 + S[y];
            }
            __syncthreads();
        } else {
        if (p1) {
#pragma unroll
            for (int yp = 0; yp < kYD; yp++) {
                double t = 0.0;
#pragma unroll
                for (int y = 0; y < kYD; y++) t = __builtin_fma(md[y > yp ? y - yp : yp - y], S[y], t);
                tl[(size_t)(yp * kXD + gx) * NC + cc] = t;
            }
        }
        PXSOM_PHASE(10);
        __syncthreads();
        // pass 2: thread <-> (y', cc), in place
        if (tid < kYD * NC) {
            const int yp = tid / NC, c2 = tid - yp * NC;
            double *col = tl + (size_t)(yp * kXD) * NC + c2;
            double Tx[kXD];
#pragma unroll
            for (int gx2 = 0; gx2 < kXD; gx2++) Tx[gx2] = col[(size_t)gx2 * NC];
#pragma unroll
            for (int xp = 0; xp < kXD; xp++) {
                double t = 0.0;
#pragma unroll
                for (int gx2 = 0; gx2 < kXD; gx2++) t = __builtin_fma(md[gx2 > xp ? gx2 - xp : xp - gx2], Tx[gx2], t);
                col[(size_t)xp * NC] = t;
            }
        }
        PXSOM_PHASE(11);
        __syncthreads();
        }
    } else {
#pragma unroll
        for (int i = 0; i < CPL; i++) {
            const int ch = nq * CPL + i;
            wv_[i] = sa.w_in[(has_node && ch < c) ? (size_t)node * c + ch : 0];
        }
        load_centring();
        if (blk < nblocks) load_rows(blk);
        PXSOM_PHASE(9);
        for (int e = tid; e < (kK * cs + kK) * ncopies; e += kStepThreads) ls[e] = 0.0;
        if (tid == 0) {
            hdr->q_n = 0u;
            hdr->bad = 0;
        }
        if (tid < 64) bias_l[6 * 64 + tid] = f32x4{kNegBig, kNegBig, kNegBig, kNegBig};   // rows of the last block without a node
        if (sa.stats_zero) {
            const int per = (sa.zero_count + (int)gridDim.x - 1) / (int)gridDim.x;
            const int z1 = min(((int)blockIdx.x + 1) * per, sa.zero_count);
            for (int e = (int)blockIdx.x * per + tid; e < z1; e += kStepThreads) sa.stats_zero[e] = 0.0;
        }
        park_centring();
        __syncthreads();
    }

    // ---- P3: new node values in registers; norms, duplicate key, maxima -----------------------------------
    double nrm = 0.0, mymax = 0.0;   // of the centred node
    float mud32[CPL];
#pragma unroll
    for (int i = 0; i < CPL; i++) mud32[i] = mu_l[nq * CPL + i < 32 ? nq * CPL + i : 0];
    const float mu_norm = mu_l[32];
    unsigned long long kkey = 0;
    {
        bool bad = false;
        if (has_node) {
#pragma clang fp contract(off)
            double den = 0.0, gain = -1.0;
            const int xp = node / kYD, yp = node - xp * kYD;
            const double *nrow = tl + (size_t)(yp * kXD + xp) * NC;
            double inv = 0.0;
            if (bmu_only) {
                gain = tl[node];
                inv = tl[kK + node];
            } else if (sa.has_update) {
                den = nrow[c];
                // gain = 1 - (1-alpha)^den (batch_gain); == 1 exactly for wide windows: the node is
                // then the window mean itself (orc_batch_update)
                if (den > 0.0) {
                    gain = batch_gain(den, sa.q, sa.sat);
                    inv = 1.0 / den;
                }
            }
#pragma unroll
            for (int i = 0; i < CPL; i++) {
                const int ch = nq * CPL + i;
                if (ch < c) {
                    double v = wv_[i];
                    if (gain >= 0.0) {   // (idle channel slots hold whatever word 0 held: never used)
                        const double num = bmu_only ? 0.0 + sdir[i] : nrow[ch];
                        v = gain == 1.0 ? num * inv : v + gain * (num * inv - v);
                    }
                    if constexpr (EXCH)
                        if (xch_late) v = __builtin_nan("");   // (the statistics it would have applied are not all there)
                    wv_[i] = v;
                    wt[(size_t)ch * kK + node] = v;
                    if (blockIdx.x == 0 && sa.w_out && (sa.has_update || sa.w_out != sa.w_in)) sa.w_out[(size_t)node * c + ch] = v;
                    bad |= !(fabs(v) <= DBL_MAX);
                    const double vc = v - (double)mud32[i];
                    nrm += vc * vc;
                    mymax = fmax(mymax, fabs(vc));
                    // duplicate key: the bit pattern rotated by a channel-dependent amount, xor-ed up (no multiplies:
                    // integer multiplies run at quarter rate; a key match is verified channel by channel anyway)
                    if constexpr (!BMU) {   // (BMU-only steps do not look for duplicates: no key)
                        const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
                        const int rot = (7 * ch + 1) & 63;
                        kkey ^= (bits << rot) | (bits >> ((64 - rot) & 63));
                    }
                }
            }
        }
        // the 4 lanes of a node are adjacent: butterfly over the quad (same order for every node, so bit-identical
        // rows get bit-identical norms and keys)
        nrm += __shfl_xor(nrm, 1);
        nrm += __shfl_xor(nrm, 2);
        if constexpr (!BMU) {
            kkey ^= __shfl_xor(kkey, 1);
            kkey ^= __shfl_xor(kkey, 2);
            if (has_node && nq == 0) key[node] = kkey;
        }
        if (bad) hdr->bad = 1;   // NaN / Inf in the codebook: every row takes the exact path
        // the two maxima only steer the scale (the exponent of the largest magnitude) and the norm bound (rounded up by a hair
        // below): binary32 roundings of them, reduced as unsigned integers -- non-negative binary32 numbers order like their bit
        // patterns, infinities included --, a third of the instructions of two binary64 reductions.  (A magnitude that rounds up
        // to the next power of two halves the scale: the scaled maximum then lies just below 128 instead of just below 256.)
        const unsigned wbits = ~pxsom::wave_min_u32(~__float_as_uint((float)mymax));
        const unsigned nbits = ~pxsom::wave_min_u32(~__float_as_uint((has_node && nrm == nrm) ? (float)nrm : 0.f));
        if (lane == 0) {
            red[wv] = (double)__uint_as_float(wbits);
            red[kStepWaves + wv] = (double)__uint_as_float(nbits) * (1.0 + 0x1p-23);
        }
    }
    PXSOM_PHASE(12);
    __syncthreads();

    // ---- P4: scale, fragments, duplicates, bias -- one phase: every thread knows its node's values, norm and key
    float fscale, wn_max;
    bool force_exact;
    {
        double maxabs = red[0], wn2max = red[kStepWaves];
#pragma unroll
        for (int i = 1; i < kStepWaves; i++) {
            maxabs = fmax(maxabs, red[i]);
            wn2max = fmax(wn2max, red[kStepWaves + i]);
        }
        // scale = 2^e with maxabs*scale in [128, 256) (pxsom_prep.h) -- but a codebook that has (nearly) collapsed onto the
        // centring vector must not blow the scale up (the rows lie where they did): at most 2^6 over what the norm of the
        // centring vector itself would choose (no reduction of the step's own for it)
        int e = 0;
        if (maxabs > 0.0 && maxabs <= DBL_MAX) {
            int ex;
            frexp(maxabs, &ex);
            e = 8 - ex;
            if (mu_norm > 0.f) {
                int exn;
                frexpf(mu_norm, &exn);
                if (e > 8 - exn + 6) e = 8 - exn + 6;
            }
            if (e > 100) e = 100;
            if (e < -100) e = -100;
        }
        const double scale = ldexp(1.0, e);
        const bool badw = hdr->bad != 0 || !(wn2max * scale * scale <= 1.0e30) || !(maxabs * scale < 256.0);   // (256: filter_cut_abs)
        fscale = (float)scale;
        // rounded up by a hair; NaN / Inf / huge codebook: every row takes the exact path
        wn_max = badw ? 0.f : (float)(sqrt(wn2max) * scale * (1.0 + 1e-6));
        force_exact = badw;
        if (has_node) {
            // A-fragment element of (node, nq): lane (nq << 4 | m) of node block b; the last block's (q, r) grid is
            // transposed (node_of_row): node 96 + t sits in row m = 4 t
            const int b = node < 96 ? node >> 4 : 6, m = node < 96 ? node & 15 : 4 * (node - 96);
            half8 fhi, flo;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                float W = 0.f;
                if (i < CPL && nq * CPL + i < c) W = (float)((wv_[i < CPL ? i : 0] - (double)mud32[i < CPL ? i : 0]) * scale);
                const _Float16 hi = (_Float16)W;
                fhi[i] = hi;
                flo[i] = (_Float16)(W - (float)hi);
            }
            const int lf = (nq << 4) | m;
            frag_l[(b * 2 + 0) * 64 + lf] = fhi;
            frag_l[(b * 2 + 1) * 64 + lf] = flo;
            // Exact duplicates of an EARLIER node are masked out of the filter (pxsom_prep.h).  The node's 4 lanes scan
            // the earlier nodes' keys together (lane nq takes prev = nq, nq + 4, ...; no early exit, so the LDS reads
            // pipeline), the smallest match is then compared channel by channel.
            auto same_as = [&](int prev) {   // all channels equal, decided by the node's 4 lanes together
                bool eq = true;
#pragma unroll
                for (int i = 0; i < CPL; i++) {
                    const int ch = nq * CPL + i;
                    if (ch < c) eq &= wt[(size_t)ch * kK + prev] == wv_[i];
                }
                int ee = eq ? 1 : 0;
                ee &= __shfl_xor(ee, 1);
                ee &= __shfl_xor(ee, 2);
                return ee != 0;
            };
            // (BMU-only steps do not look for duplicates: the first of two equal nodes takes all their rows and moves away
            // within a step; until then their rows are listed and settled exactly -- the labels do not depend on the mask)
            int hit = 0x7fffffff;
#pragma unroll 5
            for (int prev = nq; prev < (bmu_only ? 0 : kK); prev += 4) hit = min(hit, (prev < node && key[prev] == kkey) ? prev : 0x7fffffff);
            hit = min(hit, __shfl_xor(hit, 1));
            hit = min(hit, __shfl_xor(hit, 2));
            bool dup = false;
            if (hit != 0x7fffffff) {
                dup = same_as(hit);
                for (int prev = hit + 1; prev < node && !dup; prev++)   // a key collision: keep looking
                    if (key[prev] == kkey) dup = same_as(prev);
            }
            // bias of accumulator row m = 4 qf + rr of block b, replicated over the 16 pixel lanes: this lane writes 4
            const float bvv = dup ? kNegBig : (float)(-0.5 * nrm * scale * scale);
            float *bf = reinterpret_cast<float *>(bias_l) + ((size_t)(b * 64 + (m >> 2) * 16 + nq * 4) * 4 + (m & 3));
#pragma unroll
            for (int u = 0; u < 4; u++) bf[u * 4] = bvv;
        } else if (tid < 4 * kK + 48) {
            // the 12 rows of the last block that hold no node (their bias was set to kNegBig at the start)
            const int i = (tid - 4 * kK) >> 2, m = (i / 3) * 4 + (i % 3) + 1;
            const half8 z = {(_Float16)0, (_Float16)0, (_Float16)0, (_Float16)0, (_Float16)0, (_Float16)0, (_Float16)0, (_Float16)0};
            frag_l[(6 * 2 + 0) * 64 + ((nq << 4) | m)] = z;
            frag_l[(6 * 2 + 1) * 64 + ((nq << 4) | m)] = z;
        }
    }
    PXSOM_PHASE(13);
    PXSOM_PHASE(14);
    PXSOM_PHASE(15);
    __syncthreads();
    PXSOM_PHASE(22);

    // ---- P7: BMU search of this workgroup's rows -------------------------------------------------------------
    const float tol_rel = sa.tol_rel, tol_abs = sa.tol_abs, x_limit = 60000.0f;
    double *qrows = tl;   // the window-sum scratch is free now: [kQueueRows][c]
    // Round 6: a score carries the 5 bits (b, r) of its place in the lane; WHICH of the four lanes of a pixel held the winner
    // travels beside the scores through the merge (a compare + select per level) instead of in two more mantissa bits: the
    // packing term of the tolerance is 2^-18 instead of 2^-16 -- its largest term -- and a crowded codebook lists 2.5 x fewer rows
    constexpr unsigned idx_mask = 31u;
    float mus[NP][2];   // the centring vector at this lane's channels, scaled (a binary32 value times a power of two: exact)
#pragma unroll
    for (int p = 0; p < NP; p++) {
        int ch = q * CPL + 2 * p;
        if (ch > c - 2) ch = c - 2;
        mus[p][0] = mu_l[ch] * fscale;
        mus[p][1] = mu_l[ch + 1] * fscale;
    }
    if constexpr (EXCH)
        if (xch_late) blk = nblocks;   // (a poisoned codebook would send every row to the exact path: the pass is repeated anyway)
    for (; blk < nblocks; blk += gridDim.x) {
        half8 bh[TPW], bl[TPW];
        float ss[TPW];
#pragma unroll
        for (int t = 0; t < TPW; t++) {
            float acc2 = 0.f;
#pragma unroll
            for (int p = 0; p < 4; p++) {
                half2_t h2 = {(_Float16)0, (_Float16)0}, l2 = {(_Float16)0, (_Float16)0};
                if (p < NP) {
                    // x' = fl(x * scale - mu_s): one rounding (binary64 rows: formed in binary64, then rounded once more)
                    float xs0, xs1;
                    if constexpr (sizeof(T) == 8) {
                        xs0 = (float)__builtin_fma((double)raw[t][p < NP ? p : 0].x, (double)fscale, -(double)mus[p < NP ? p : 0][0]);
                        xs1 = (float)__builtin_fma((double)raw[t][p < NP ? p : 0].y, (double)fscale, -(double)mus[p < NP ? p : 0][1]);
                    } else {
                        xs0 = fmaf((float)raw[t][p < NP ? p : 0].x, fscale, -mus[p < NP ? p : 0][0]);
                        xs1 = fmaf((float)raw[t][p < NP ? p : 0].y, fscale, -mus[p < NP ? p : 0][1]);
                    }
                    h2[0] = (_Float16)xs0;
                    h2[1] = (_Float16)xs1;
                    l2[0] = (_Float16)(xs0 - (float)h2[0]);
                    l2[1] = (_Float16)(xs1 - (float)h2[1]);
                    acc2 = __builtin_amdgcn_fdot2(h2, h2, acc2, false);
                }
                bh[t][2 * p] = h2[0];
                bh[t][2 * p + 1] = h2[1];
                bl[t][2 * p] = l2[0];
                bl[t][2 * p + 1] = l2[1];
            }
            ss[t] = acc2;
        }
        float m1[TPW], m2[TPW];
#pragma unroll
        for (int t = 0; t < TPW; t++) m1[t] = m2[t] = kNegBig;
        PXSOM_PHASE(23);
#pragma unroll
        for (int b = 0; b < kNB; b++) {
            const half8 wa0 = frag_l[(b * 2 + 0) * 64 + lane], wa1 = frag_l[(b * 2 + 1) * 64 + lane];
            const f32x4 bb = bias_l[b * 64 + lane];
            f32x4 acc[TPW];
            // Wh*Xh + Wh*Xl + Wl*Xh, the tiles' chains interleaved
#pragma unroll
            for (int t = 0; t < TPW; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa0, bh[t], bb, 0, 0, 0);
#pragma unroll
            for (int t = 0; t < TPW; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa0, bl[t], acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < TPW; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa1, bh[t], acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < TPW; t++) {
                if (b < kNB - 1) {
                    top2_quad(m1[t], m2[t], pack_idx(acc[t][0], (unsigned)(b * 4 + 0), idx_mask),
                              pack_idx(acc[t][1], (unsigned)(b * 4 + 1), idx_mask),
                              pack_idx(acc[t][2], (unsigned)(b * 4 + 2), idx_mask),
                              pack_idx(acc[t][3], (unsigned)(b * 4 + 3), idx_mask));
                } else {   // last block: only accumulator register 0 holds real nodes (K = 100)
                    const float p0 = pack_idx(acc[t][0], (unsigned)(b * 4 + 0), idx_mask);
                    m2[t] = __builtin_amdgcn_fmed3f(m1[t], m2[t], p0);
                    m1[t] = max_of(m1[t], p0);
                }
            }
        }
        PXSOM_PHASE(24);
        // the next block's rows (usually none: one block per workgroup) -- the current values are kept
        P2 cur[TPW][NP];
#pragma unroll
        for (int t = 0; t < TPW; t++)
#pragma unroll
            for (int p = 0; p < NP; p++) cur[t][p] = raw[t][p];
        if (blk + gridDim.x < nblocks) load_rows(blk + gridDim.x);
#pragma unroll
        for (int t = 0; t < TPW; t++) {
            // merge of the 4 lane groups that share a pixel: afterwards all four hold the pixel's top-2
            // (v_permlane16/32_swap of a value with itself hands BOTH lanes of a pair {the even lane row's, the odd one's} resp.
            // {the lower half's, the upper half's} in that order: the comparison below gives both the same answer)
            float a1 = m1[t], a2 = m2[t], s2 = ss[t];
            unsigned wq;   // lane group (0..3) that held the winner
            {
                const F2 e1 = xchg16(a1), e2 = xchg16(a2), es = xchg16(s2);
                wq = e1.a >= e1.b ? 0u : 1u;
                a1 = max_of(e1.a, e1.b);
                a2 = fmaxf(fmaxf(min_of(e1.a, e1.b), e2.a), e2.b);
                s2 = es.a + es.b;
            }
            {
                const F2 e1 = xchg32(a1), e2 = xchg32(a2), es = xchg32(s2);
                const unsigned mine = wq | ((unsigned)q & 2u);   // (the upper half's lane groups are 2 and 3)
                const auto eq = __builtin_amdgcn_permlane32_swap(mine, mine, false, false);
                wq = e1.a >= e1.b ? eq[0] : eq[1];
                a1 = max_of(e1.a, e1.b);
                a2 = fmaxf(fmaxf(min_of(e1.a, e1.b), e2.a), e2.b);
                s2 = es.a + es.b;
            }
            // |Xh| <= |X| (1 + 2^-11): folded into the 1.001 factor with the sqrt's ulp
            const float xn = __builtin_amdgcn_sqrtf(s2) * 1.001f;
            const float tol = tol_rel * (xn * wn_max + 0.5f * wn_max * wn_max) + tol_abs * (xn + wn_max) + kTolFloor;
            const unsigned nonfinite = (unsigned)((__float_as_uint(s2) & 0x7f800000u) == 0x7f800000u);
            const int64_t row = blk * kRowsPerWg + (int64_t)wv * kRowsPerWave + t * 16 + pix;
            const bool valid = row < n;
            const bool amb = valid && ((!((a1 - a2) > tol)) || !(xn < x_limit) || nonfinite != 0u || force_exact);
            // id (q, b, r) -> node: 16 b + 4 q + r, the last block's 4x4 (q, r) grid transposed (node_of_row)
            const unsigned id = __float_as_uint(a1) & idx_mask;
            const unsigned wb = id >> 2, wr = id & 3u;
            const unsigned real = wb == (unsigned)(kNB - 1) ? 16u * wb + 4u * wr + wq : 16u * wb + 4u * wq + wr;
            if (valid && !amb) {
                double *tab = ls + (size_t)(pix & (ncopies - 1)) * tstride;
                double *dst = tab + (size_t)real * cs + q * CPL;
#pragma unroll
                for (int p = 0; p < NP; p++) {
                    if (q * CPL + 2 * p <= c - 2) {   // clamped slots re-read the last pair: not theirs
                        double v0 = (double)cur[t][p].x, v1 = (double)cur[t][p].y;
                        if constexpr (sizeof(T) == 8) {   // binary64 rows: rounded to the run's quantum (exact, order-free sums)
                            v0 = qround(v0, qmagic);
                            v1 = qround(v1, qmagic);
                        }
                        __hip_atomic_fetch_add(dst + 2 * p, v0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_add(dst + 2 * p + 1, v1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
                if (q == 0)
                    __hip_atomic_fetch_add(tab + (size_t)kK * cs + real, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            // listed rows: values into the queue (all four lanes of a pixel agree on amb and on the slot)
            const unsigned mask16 = (unsigned)(__ballot(amb) & 0xffffull);
            if (mask16) {
                unsigned base = 0;
                if (lane == 0) base = atomicAdd(&hdr->q_n, (unsigned)__popc(mask16));
                base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
                const unsigned pos = base + (unsigned)__popc(mask16 & ((1u << pix) - 1u));
                if (amb && pos < (unsigned)kQueueRows) {
#pragma unroll
                    for (int p = 0; p < NP; p++) {
                        if (q * CPL + 2 * p <= c - 2) {
                            qrows[(size_t)pos * c + q * CPL + 2 * p] = (double)cur[t][p].x;
                            qrows[(size_t)pos * c + q * CPL + 2 * p + 1] = (double)cur[t][p].y;
                        }
                    }
                }
                // queue full: the rows that did not fit are settled on the spot, one at a time, through this wave's slot
                unsigned late = (unsigned)(__ballot(amb && pos >= (unsigned)kQueueRows) & 0xffffull);
                while (late) {
                    const int src = __builtin_ctz(late);
                    late &= late - 1u;
                    double *slot = ovf + (size_t)wv * 32;
                    if (pix == src) {
#pragma unroll
                        for (int p = 0; p < NP; p++) {
                            if (q * CPL + 2 * p <= c - 2) {
                                slot[q * CPL + 2 * p] = (double)cur[t][p].x;
                                slot[q * CPL + 2 * p + 1] = (double)cur[t][p].y;
                            }
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    exact_row_from_lds(slot, c, wt, ls + (size_t)(wv & (ncopies - 1)) * tstride, lane, qmagic, cs);
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
    }
    // the listed rows of ALL rounds are settled together, by whichever wave is free: a round no longer ends in two barriers
    // (the queue keeps the rows' values; what does not fit was settled on the spot)
    PXSOM_PHASE(16);
    __syncthreads();   // every wave is through its tiles: the queue is complete
    {
        const unsigned queued = hdr->q_n < (unsigned)kQueueRows ? hdr->q_n : (unsigned)kQueueRows;
#ifdef PXSOM_STEP_COUNT_LISTED
        if (tid == 0) {
            const unsigned slot = *reinterpret_cast<volatile unsigned *>(&g_dbg_launch) & 4095u;
            atomicAdd(&g_dbg_listed[slot], hdr->q_n);
            atomicMax(&g_dbg_listed_max[slot], hdr->q_n);
            if (blockIdx.x == 0) g_dbg_rows[slot] = (unsigned)n;
        }
#endif
        for (unsigned i = wv; i < queued; i += kStepWaves)
            exact_row_from_lds(qrows + (size_t)i * c, c, wt, ls + (size_t)(wv & (ncopies - 1)) * tstride, lane, qmagic, cs);
        __syncthreads();
        PXSOM_PHASE(17);
#ifdef PXSOM_PHASE_TIMING
        if (tid == 0 && blockIdx.x == 0) g_phase_ticks[21] = (long long)queued;
#endif
    }
    // ---- P9: flush.  Every workgroup starts at a different WORD (blockIdx * 97 mod the table's length), so that the workgroups
    // of a launch -- which all get here at about the same time -- are spread over the whole buffer instead of queueing up on the
    // same few cache lines (five starting points, as it was until round 4, left 14 - 25 workgroups of a small step on each:
    // scripts/ubench/flush_replicas.hip issues 69 x 2 300 atomics in under 1 us this way)
    // (Round 6: the division per element is not what this phase waits for -- with one division per thread and the LDS reads in
    // front of the atomics it takes 2.0 us on a tail step against 1.7: 69 workgroups x ~1 650 binary64 atomics on the same 144
    // cache lines are served by the memory-side atomic unit at its own pace; profiles/r06/step_phase_timing.txt)
    {
        const int total = kK * c + kK;
        const int shift = (int)((blockIdx.x * 97u) % (unsigned)total);
        for (int e0 = tid; e0 < total; e0 += kStepThreads) {
            int e = e0 + shift;
            if (e >= total) e -= total;
            const int node = e / c;                                              // e -> (node, channel) | count
            const int le = e < kK * c ? node * cs + (e - node * c) : kK * cs + (e - kK * c);
            double v = ls[le];
            for (int j = 1; j < ncopies; j++) v += ls[le + (size_t)j * tstride];
            if (v != 0.0) __hip_atomic_fetch_add(stats + e, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    PXSOM_PHASE(18);
    if constexpr (EXCH) {
        if (sa.xch_signal) {
            // the last workgroup through the flush hands this rank's statistics to every rank
            __shared__ int s_xch_last;
            __threadfence();
            __syncthreads();
            // the ticket counts the workgroups in its low 16 bits (a grid holds at most 2 x 256) and the late ones above them
            __shared__ int s_xch_poison;
            if (tid == 0) {
                const unsigned mine = 1u + (xch_late ? 0x10000u : 0u);
                const unsigned t = __hip_atomic_fetch_add(sa.xch_ticket, mine, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                s_xch_last = (t & 0xffffu) == gridDim.x - 1u;
                s_xch_poison = ((t + mine) >> 16) != 0u;
                if (s_xch_last) __hip_atomic_store(sa.xch_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
            if (s_xch_last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                const int parity = (int)(sa.xch_signal & 1ull), total = kK * c + kK;
                for (int e = tid; e < total; e += kStepThreads) {
                    const double v = s_xch_poison ? __builtin_nan("") : __hip_atomic_load(stats + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    for (int p = 0; p < sa.xch_nranks; p++)
                        __hip_atomic_store(pxsom::p2p_slot(sa.xch_peers[p], parity, sa.xch_rank, sa.xch_nranks, (size_t)sa.xch_max_count) + e, v,
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                __threadfence_system();
                __syncthreads();
                if (tid < sa.xch_nranks)
                    __hip_atomic_store(&reinterpret_cast<pxsom::P2PBlock *>(sa.xch_peers[tid])->flags[parity][sa.xch_rank], sa.xch_signal,
                                       __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
#ifdef PXSOM_PHASE_TIMING
    __builtin_amdgcn_s_waitcnt(0);
    PXSOM_PHASE(19);
    if (tid == 0) {
        g_block_ticks[blockIdx.x * 2] = t_start;
        g_block_ticks[blockIdx.x * 2 + 1] = wall_clock64();
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// Codebooks too big for the all-in-one step kernel (K = 400 x C = 40: fragments + statistics table exceed one CU's
// LDS; C = 100: the same): the pending update and the codebook preparation run ONCE, in one single-workgroup launch
// (instead of copy + update + 2 clears + prep), and leave what the generic BMU search reads in the assign workspace:
// fragments, bias, header, transposed binary64 copy.  Same arithmetic as the head of batch_step_kernel: separable
// window sums in registers, node values / norms / duplicate keys in the registers of thread <-> (node, part),
// duplicates by a key scan shared by the node's lanes.  1024 threads.
// ------------------------------------------------------------------------------------------------
constexpr int kUpdThreads = 1024, kUpdWaves = 16;

template <int XD, int YD, int CPP>
__global__ __launch_bounds__(kUpdThreads) void batch_update_prep_kernel(StepArgs sa, int c, AssignHdr *hdr_g, half8 *wfrag,
                                                                        f32x4 *bias_g, double *wt_out, int nb, int nch,
                                                                        int cpl, int idx_bits, int parts_log2,
                                                                        float *w32_out, int cp32, int npk)
{
    constexpr int K = XD * YD;
    extern __shared__ __attribute__((aligned(16))) char upd_smem[];
    const int NC = c + 1;
    double *tl = reinterpret_cast<double *>(upd_smem);                       // [K * NC] window sums; then W_new [K][c]
    unsigned long long *key = reinterpret_cast<unsigned long long *>(tl + (size_t)K * NC);   // [K]
    double *red = reinterpret_cast<double *>(key + K);                       // [3 * waves]
    float *biasv = reinterpret_cast<float *>(red + 3 * kUpdWaves);           // [K]
    int *flags = reinterpret_cast<int *>(biasv + K);                         // [0]: NaN / Inf met
    float *s_mu = reinterpret_cast<float *>(flags + 16);                     // [128] the run's centring vector (zeros: not centred)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int parts = 1 << parts_log2;
    const int node = tid >> parts_log2, part = tid & (parts - 1);
    const bool has_node = node < K;
    const int cpp = (c + parts - 1) / parts;          // channels per part (<= CPP)
    const int ch0 = part * cpp;
    PXSOM_PHASE(0);

    // ---- requests, in the order they are needed: statistics, old node values.  Every global access of this kernel is
    // coalesced (element e = tid + 1024 u of a row-major array): one workgroup issues all of them, and a wave-load
    // whose lanes sit 8 c bytes apart costs a cache line per lane.
    constexpr int kMaxE = (K * 128 + kUpdThreads - 1) / kUpdThreads;   // elements of W per thread (c <= 128)
    double S[YD], wold[kMaxE < 16 ? kMaxE : 16];
    constexpr int kE = kMaxE < 16 ? kMaxE : 16;
    const int ne = (K * c + kUpdThreads - 1) / kUpdThreads;            // <= kE (checked by the host)
    const bool p1 = sa.has_update && tid < XD * NC;
    const int gx = p1 ? tid / NC : 0, cc = p1 ? tid - gx * NC : 0;
    if (sa.has_update) {
#pragma unroll
        for (int y = 0; y < YD; y++)
            S[y] = sa.stats_prev[cc < c ? (size_t)(gx * YD + y) * c + cc : (size_t)K * c + gx * YD + y];
    }
#pragma unroll
    for (int u = 0; u < kE; u++) wold[u] = sa.w_in[(u < ne && tid + kUpdThreads * u < K * c) ? tid + kUpdThreads * u : 0];
    if (tid == 0) flags[0] = 0;
    if (tid < kFilterMaxChannels) s_mu[tid] = (sa.mu32 && tid < c) ? sa.mu32[tid] : 0.f;
    const float mu_norm = sa.mu32 ? sa.mu32[kFilterMaxChannels] : 0.f;
    // Several workgroups run this kernel: each redoes the update and the norms (they need all of it, and it stays in
    // their own LDS: no traffic between workgroups), and takes a share of the OUTPUT -- node blocks [b0, b1) of the
    // duplicate test, fragments, bias and the two codebook copies; workgroup 0 also writes W_g and the header.
    const int bper = (nb + (int)gridDim.x - 1) / (int)gridDim.x;
    const int b0 = (int)blockIdx.x * bper, b1 = min(b0 + bper, nb);
    const int n0 = min(16 * b0, K), n1 = min(16 * b1, K);
    if (sa.stats_zero) {
        const int zper = (sa.zero_count + (int)gridDim.x - 1) / (int)gridDim.x;
        const int z1 = min(((int)blockIdx.x + 1) * zper, sa.zero_count);
        for (int e = (int)blockIdx.x * zper + tid; e < z1; e += kUpdThreads) sa.stats_zero[e] = 0.0;
    }
    PXSOM_PHASE(1);
    if (sa.has_update) {
        const double thr = sa.thr;
        const int r = thr < 0.0 ? -1 : (thr > 1.0e6 ? 1000000 : (int)floor(thr));
        // window mask as a uniform select per term (|y - y'| <= r ? 1 : 0): no table of masks to keep in registers
        if (p1) {
#pragma unroll
            for (int yp = 0; yp < YD; yp++) {
                double t = 0.0;
#pragma unroll
                for (int y = 0; y < YD; y++) t = __builtin_fma((y > yp ? y - yp : yp - y) <= r ? 1.0 : 0.0, S[y], t);
                tl[(size_t)(yp * XD + gx) * NC + cc] = t;
            }
        }
        PXSOM_PHASE(2);
        __syncthreads();
        if (tid < YD * NC) {
            const int yp = tid / NC, c2 = tid - yp * NC;
            double *col = tl + (size_t)(yp * XD) * NC + c2;
            double Tx[XD];
#pragma unroll
            for (int gx2 = 0; gx2 < XD; gx2++) Tx[gx2] = col[(size_t)gx2 * NC];
#pragma unroll
            for (int xp = 0; xp < XD; xp++) {
                double t = 0.0;
#pragma unroll
                for (int gx2 = 0; gx2 < XD; gx2++) t = __builtin_fma((gx2 > xp ? gx2 - xp : xp - gx2) <= r ? 1.0 : 0.0, Tx[gx2], t);
                col[(size_t)xp * NC] = t;
            }
        }
        PXSOM_PHASE(3);
        __syncthreads();
        // gain and 1/den of every node, once (key[] / biasv[] double as scratch until they get their own contents)
        double *gain_l = reinterpret_cast<double *>(key);        // [K]
        if (tid < K) {
            const int xp = tid / YD, yp = tid - xp * YD;
            const double den = tl[(size_t)(yp * XD + xp) * NC + c];
            gain_l[tid] = den > 0.0 ? batch_gain(den, sa.q, sa.sat) : -1.0;
            tl[(size_t)(yp * XD + xp) * NC + c] = den > 0.0 ? 1.0 / den : 0.0;     // the count column now holds 1/den
        }
        PXSOM_PHASE(4);
        __syncthreads();
        // new node values, element-wise (coalesced); kept in registers until every read of the window sums is done
        {
#pragma clang fp contract(off)
            int nd = tid / c, j = tid - nd * c;
            const int dnd = kUpdThreads / c, dj = kUpdThreads % c;
#pragma unroll
            for (int u = 0; u < kE; u++) {
                const int e = tid + kUpdThreads * u;
                if (u < ne && e < K * c) {
                    const int xp = nd / YD, yp = nd - xp * YD;
                    const double *nrow = tl + (size_t)(yp * XD + xp) * NC;
                    const double gain = gain_l[nd];
                    double v = wold[u];
                    if (gain >= 0.0) {
                        const double mean = nrow[j] * nrow[c];
                        v = gain == 1.0 ? mean : v + gain * (mean - v);
                    }
                    wold[u] = v;
                    if (sa.w_out && blockIdx.x == 0) sa.w_out[e] = v;
                }
                nd += dnd;
                j += dj;
                if (j >= c) {
                    j -= c;
                    nd++;
                }
            }
        }
        PXSOM_PHASE(5);
        __syncthreads();
    }
    // the window-sum region becomes W_new [K][c]
#pragma unroll
    for (int u = 0; u < kE; u++)
        if (u < ne && tid + kUpdThreads * u < K * c) tl[tid + kUpdThreads * u] = wold[u];
    __syncthreads();

    // ---- norms, keys, maxima: thread <-> (node, part) over the LDS copy
    double nrm = 0.0, raw2 = 0.0, mymax = 0.0, wv_[CPP];   // (nrm, mymax: of the centred node)
    unsigned long long kkey = 0;
    bool bad = false;
#pragma unroll
    for (int i = 0; i < CPP; i++) {
        const int ch = ch0 + i;
        wv_[i] = 0.0;
        if (has_node && i < cpp && ch < c) {
            const double v = tl[(size_t)node * c + ch];
            wv_[i] = v;
            bad |= !(fabs(v) <= DBL_MAX);
            const double vc = v - (double)s_mu[ch < kFilterMaxChannels ? ch : 0];
            nrm += vc * vc;
            raw2 += v * v;
            mymax = fmax(mymax, fabs(vc));
            const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
            const int rot = (7 * ch + 1) & 63;
            kkey ^= (bits << rot) | (bits >> ((64 - rot) & 63));
        }
    }
    for (int m = 1; m < parts; m <<= 1) {   // the lanes of a node are adjacent: butterfly
        nrm += __shfl_xor(nrm, m);
        raw2 += __shfl_xor(raw2, m);
        kkey ^= __shfl_xor(kkey, m);
    }
    PXSOM_PHASE(6);
    __syncthreads();   // (key[] was scratch for the gains)
    if (has_node && part == 0) key[node] = kkey;
    if (bad) flags[0] = 1;
    {
        const double wmax = -pxsom::wave_min_f64(-mymax);
        const double nmax = -pxsom::wave_min_f64(-((has_node && nrm == nrm) ? nrm : 0.0));
        const double rmax = -pxsom::wave_min_f64(-((has_node && raw2 == raw2) ? raw2 : 0.0));
        if (lane == 0) {
            red[wv] = wmax;
            red[kUpdWaves + wv] = nmax;
            red[2 * kUpdWaves + wv] = rmax;
        }
    }
    __syncthreads();
    double maxabs = red[0], wn2max = red[kUpdWaves], raw2max = red[2 * kUpdWaves];
#pragma unroll
    for (int i = 1; i < kUpdWaves; i++) {
        maxabs = fmax(maxabs, red[i]);
        wn2max = fmax(wn2max, red[kUpdWaves + i]);
        raw2max = fmax(raw2max, red[2 * kUpdWaves + i]);
    }
    int e = 0;
    if (maxabs > 0.0 && maxabs <= DBL_MAX) {
        int ex;
        frexp(maxabs, &ex);
        e = 8 - ex;
        if (mu_norm > 0.f) {   // a codebook collapsed onto the centring vector must not blow the scale up (batch_step_kernel P4)
            int exn;
            frexpf(mu_norm, &exn);
            if (e > 8 - exn + 6) e = 8 - exn + 6;
        }
        if (e > 100) e = 100;
        if (e < -100) e = -100;
    }
    const double scale = ldexp(1.0, e);
    __syncthreads();
    const bool badw = flags[0] != 0 || !(wn2max * scale * scale <= 1.0e30) || !(maxabs * scale < 256.0);   // (256: filter_cut_abs)
    if (tid == 0 && blockIdx.x == 0) {
        hdr_g->amb_count = 0;
        hdr_g->scale = (float)scale;
        hdr_g->wn_max = badw ? 0.f : (float)(sqrt(wn2max) * scale * (1.0 + 1e-6));
        hdr_g->force_exact = badw ? 1 : 0;
        hdr_g->tol_rel = sa.tol_rel;
        hdr_g->tol_rel_coarse = sa.tol_rel + 2.5f * 0x1.004p-10f;   // (the register-resident filter's first stage; not used by the shapes this kernel serves)
        hdr_g->tol_abs = sa.tol_abs;
        hdr_g->x_limit = 60000.0f;
        hdr_g->nb = nb;
        hdr_g->nch = nch;
        hdr_g->cpl = cpl;
        hdr_g->idx_bits = idx_bits;
        hdr_g->node_bits = idx_bits;
        // (the screened exact kernel's rounding bound is on the raw vectors)
        hdr_g->wn_raw = badw ? 0.f : (float)(sqrt(raw2max) * (1.0 + 1e-6));
        // a row the filter vouches for has |x_j * scale| < x_limit + max_j |mu_s_j| (pxsom_prep.h)
        double mumax = 0.0;
        for (int j = 0; j < (c < kFilterMaxChannels ? c : kFilterMaxChannels); j++) mumax = fmax(mumax, fabs((double)s_mu[j]) * scale);
        int t = 0;
        while (t < 60 && !(60000.0 + mumax <= ldexp(65536.0, t))) t++;
        hdr_g->fix_exp = e - t;
        hdr_g->centred = sa.mu32 ? 1 : 0;
    }
    if (blockIdx.x == 0 && tid < kFilterMaxChannels) hdr_g->mu_s[tid] = (float)((double)s_mu[tid] * scale);
    PXSOM_PHASE(7);
    // ---- exact duplicates of an earlier node: key scan shared by the node's lanes, then channel-by-channel
    if (has_node && node >= n0 && node < n1) {
        auto same_as = [&](int prev) {
            bool eq = true;
#pragma unroll
            for (int i = 0; i < CPP; i++)
                if (i < cpp && ch0 + i < c) eq &= tl[(size_t)prev * c + ch0 + i] == wv_[i];
            int ee = eq ? 1 : 0;
            for (int m = 1; m < parts; m <<= 1) ee &= __shfl_xor(ee, m);
            return ee != 0;
        };
        int hit = 0x7fffffff;
#pragma unroll 4
        for (int prev = part; prev < K; prev += parts) hit = min(hit, (prev < node && key[prev] == kkey) ? prev : 0x7fffffff);
        for (int m = 1; m < parts; m <<= 1) hit = min(hit, __shfl_xor(hit, m));
        bool dup = false;
        if (hit != 0x7fffffff) {
            dup = same_as(hit);
            for (int prev = hit + 1; prev < node && !dup; prev++)
                if (key[prev] == kkey) dup = same_as(prev);
        }
        if (part == 0) biasv[node] = dup ? kNegBig : (float)(-0.5 * nrm * scale * scale);
    }
    PXSOM_PHASE(8);
    __syncthreads();
    // ---- what the generic BMU search reads: fragments, bias, transposed copy (pxsom_prep.h layouts)
    if (npk > 0) {   // packed K axis (pxsom_assign.h packed_k; pxsom_prep.h holds the same loop)
        const int g8 = c / 8;
        for (int f = tid; f < (b1 - b0) * npk * 64; f += kUpdThreads) {
            const int fl = f & 63, m = (f >> 6) % npk, b = b0 + (f >> 6) / npk;
            const int nd = node_of_row(b, fl & 15, nb);
            const int sg = 4 * m + (fl >> 4), term = sg / g8, ch0 = 8 * (sg - term * g8);
            half8 fr;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                float W = 0.f;
                if (term < 2 && nd < K) W = (float)(tl[(size_t)nd * c + ch0 + i] * scale);   // (packed K: binary16 rows, never centred)
                const _Float16 hi = (_Float16)W;
                fr[i] = term == 0 ? hi : (_Float16)(W - (float)hi);
            }
            wfrag[(size_t)(b * npk + m) * 64 + fl] = fr;
        }
    }
    const int nsteps = 2 * nch;
    for (int f = tid; f < (npk > 0 ? 0 : (b1 - b0) * nch * 64); f += kUpdThreads) {
        const int fl = f & 63, h = (f >> 6) % nch, b = b0 + (f >> 6) / nch;
        const int m = fl & 15, q = fl >> 4;
        const int nd = node_of_row(b, m, nb);
        half8 fhi, flo;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int ch = h * 4 * cpl + q * cpl + i;
            float W = 0.f;
            if (i < cpl && ch < c && nd < K) W = (float)((tl[(size_t)nd * c + ch] - (double)s_mu[ch < kFilterMaxChannels ? ch : 0]) * scale);
            const _Float16 hi = (_Float16)W;
            fhi[i] = hi;
            flo[i] = (_Float16)(W - (float)hi);
        }
        wfrag[(size_t)(b * nsteps + 2 * h) * 64 + fl] = fhi;
        wfrag[(size_t)(b * nsteps + 2 * h + 1) * 64 + fl] = flo;
    }
    for (int f = tid; f < (b1 - b0) * 64; f += kUpdThreads) {
        const int fl = f & 63, b = b0 + (f >> 6), q = fl >> 4;
        f32x4 bv;
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            const int nd = node_of_row(b, q * 4 + rr, nb);
            bv[rr] = nd < K ? biasv[nd] : kNegBig;
        }
        bias_g[(size_t)b * 64 + fl] = bv;
    }
    PXSOM_PHASE(9);
    {   // binary32 copy for the long-list exact kernel (rows zero-padded to cp32 channels)
        const int first = n0 * cp32 + tid;
        int nd = first / cp32, j = first - nd * cp32;
        const int dn = kUpdThreads / cp32, dj = kUpdThreads % cp32;
        for (int e2 = first; e2 < n1 * cp32; e2 += kUpdThreads) {
            w32_out[e2] = j < c ? (float)tl[(size_t)nd * c + j] : 0.f;
            nd += dn;
            j += dj;
            if (j >= cp32) {
                j -= cp32;
                nd++;
            }
        }
    }
    if (wt_out && n1 > n0) {   // [c][K]: this workgroup's nodes of every channel (runs of n1 - n0 consecutive words)
        const int span = n1 - n0;
        for (int e2 = tid; e2 < span * c; e2 += kUpdThreads) {
            const int j = e2 / span, nd = n0 + (e2 - j * span);
            wt_out[(size_t)j * K + nd] = tl[(size_t)nd * c + j];
        }
    }
    PXSOM_PHASE(10);
}

template <typename T, int CPL>
int launch_step(const T *x, int64_t n, int c, int64_t ldx, double *stats, const StepArgs &sa, int tiles_per_wave,
                hipStream_t st)
{
    const size_t lds = step_lds(c).total;
    auto k1 = batch_step_kernel<T, CPL, 1, false>;
    auto k2 = batch_step_kernel<T, CPL, 2, false>;
    auto k4 = batch_step_kernel<T, CPL, 4, false>;
    auto k1b = batch_step_kernel<T, CPL, 1, true>;
    auto k2b = batch_step_kernel<T, CPL, 2, true>;
    auto x1 = batch_step_kernel<T, CPL, 1, false, true>;     // the exchange inside the launch (sharded runs, peer-to-peer blocks)
    auto x2 = batch_step_kernel<T, CPL, 2, false, true>;
    auto x4 = batch_step_kernel<T, CPL, 4, false, true>;
    auto x1b = batch_step_kernel<T, CPL, 1, true, true>;
    auto x2b = batch_step_kernel<T, CPL, 2, true, true>;
    static pxsom::PerDevice<size_t> attr_lds_on;
    size_t &attr_lds = attr_lds_on.here();
    if (attr_lds < lds) {
        for (const void *fn : {reinterpret_cast<const void *>(k1), reinterpret_cast<const void *>(k2), reinterpret_cast<const void *>(k4),
                               reinterpret_cast<const void *>(k1b), reinterpret_cast<const void *>(k2b), reinterpret_cast<const void *>(x1),
                               reinterpret_cast<const void *>(x2), reinterpret_cast<const void *>(x4), reinterpret_cast<const void *>(x1b),
                               reinterpret_cast<const void *>(x2b)}) {
            const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess)
                return pxsom::fail(PXSOM_ERR_HIP, "batch step kernel: cannot raise the LDS limit to %zu bytes: %s", lds,
                                   hipGetErrorString(e));
        }
        attr_lds = lds;
    }
    // 16-row tiles per wave: one for the steps that fit the chip in one round (a step is latency, the shortest chain wins:
    // measured on 16 K-row steps, 0.93 ms per 64-step pass against 0.99 with two); the large steps of a schedule take 2 or
    // 4 -- a round costs ~3.5 us whatever it holds, and the search itself runs at a fraction of the filter kernel's rate
    // workgroup slots: one per CU.  (Two fit -- 77 KB of LDS, <= 128 VGPRs each -- and were measured on the two-phase
    // schedule's large steps: SLOWER, pass 0.505 -> 0.574 ms; a step's time grows with the number of workgroups that flush
    // their tables into the same 18 KB of statistics.)
    const int64_t cus = pxsom::device_cu_count(), slots = cus;
    int tpw = tiles_per_wave == 2 ? 2 : (tiles_per_wave == 4 ? 4 : 1);
    if (tiles_per_wave <= 0) {
        const int64_t blocks1 = (n + kStepWaves * 16 - 1) / (kStepWaves * 16);
        tpw = blocks1 <= slots ? 1 : (blocks1 <= 2 * slots ? 2 : 4);
        // (steps that fit one round keep ONE tile per wave: with two, half as many workgroups flush their tables -- fewer
        // atomics -- but search twice as long: pass 0.336 against 0.325 ms, profiles/r06/experiments.txt)
        // binary64 rows (the drop-in classes' tables): four tiles per wave spill 182 VGPRs, and more rounds of one tile cost
        // less than that -- measured on 1 M x 22, default schedule: pass 0.520 ms (by size) / 0.487 (two) / 0.474 (one)
        if (sizeof(T) == 8) tpw = 1;
    }
    // (the kernel's own test of the pending update's threshold, made here: a BMU-only step takes the specialised kernel)
    const bool bmu = sa.has_update != 0 && sa.thr >= 0.0 && sa.thr < 1.0;
    if (bmu && tpw == 4) tpw = 2;   // (BMU-only steps that large do not occur on the default schedule: two tiles, more rounds)
    const int64_t rows_per_wg = (int64_t)kStepWaves * 16 * tpw;
    // a step larger than one block per slot: the fewest rounds, spread evenly (853 blocks -> 214 workgroups x 4, not 256 x 3.3)
    const int64_t nblocks = std::max<int64_t>((n + rows_per_wg - 1) / rows_per_wg, 1);
    const int64_t rounds = (nblocks + slots - 1) / slots;
    const int grid = (int)((nblocks + rounds - 1) / rounds);
    auto kern = tpw == 1 ? (bmu ? k1b : k1) : (tpw == 2 ? (bmu ? k2b : k2) : k4);
    if (sa.xch_peers) kern = tpw == 1 ? (bmu ? x1b : x1) : (tpw == 2 ? (bmu ? x2b : x2) : x4);
    // row f of a scheduled step -> f / group_w by a multiplication: m = floor(2^(31 + s) / w) + 1 with 2^(s-1) < w <= 2^s is exact
    // for f < 2^31 (step_fused_shape: a step holds fewer than 2^31 rows... 2^32 by the shape test, 2^31 here)
    StepArgs sk = sa;
    {
        const unsigned w = (unsigned)std::max(sa.group_w, 1);
        int s = 0;
        while (((uint64_t)1 << s) < w) s++;
        if (s == 0) {   // w == 1: f itself
            sk.group_magic = 0x80000000u;
            sk.group_shift = 0;
            // (mulhi(f, 2^31) = f >> 1: not f -- the kernel's plain-view branch is taken for w == 1, the fields are unused)
        } else {
            sk.group_magic = (unsigned)((((uint64_t)1 << (31 + s)) / w) + 1u);
            sk.group_shift = s - 1;
        }
        if (n >= ((int64_t)1 << 31)) return pxsom::fail(PXSOM_ERR_INVALID_ARG, "batch step: %lld rows in one step (limit 2^31)", (long long)n);
    }
#ifdef PXSOM_STEP_COUNT_LISTED
    hipLaunchKernelGGL(dbg_next_launch, dim3(1), dim3(1), 0, st);
#endif
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kStepThreads), lds, st, x, n, c, ldx, stats, sk);
    PXSOM_LAUNCH_CHECK("batch_step_kernel");
    return PXSOM_OK;
}

}  // namespace

// The pending update + codebook preparation for the generic BMU search, one launch (false: shape not covered).
bool launch_update_prepare(const StepArgs &sa, int xdim, int ydim, int c, char *ws, const Layout &L, hipStream_t st, int *rc)
{
    *rc = PXSOM_OK;
    const int k = xdim * ydim;
    if (!((xdim == 10 && ydim == 10) || (xdim == 20 && ydim == 20)) || k * 1 > kUpdThreads) return false;
    int pl = 0;
    while ((k << (pl + 1)) <= kUpdThreads && pl < 3) pl++;
    const int parts = 1 << pl, cpp = (c + parts - 1) / parts;
    if (cpp > 20 || xdim * (c + 1) > kUpdThreads || (k * c + kUpdThreads - 1) / kUpdThreads > 16) return false;
    const size_t lds = ((size_t)k * (c + 1) + k + 3 * kUpdWaves) * 8 + (size_t)k * 4 + 64 + kFilterMaxChannels * 4;
    if (lds > 158 * 1024) return false;
    auto kern = xdim == 10 ? batch_update_prep_kernel<10, 10, 20> : batch_update_prep_kernel<20, 20, 20>;
    static pxsom::PerDevice<size_t> attr[2];
    size_t &have = attr[xdim == 10 ? 0 : 1].here();
    if (have < lds) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            *rc = pxsom::fail(PXSOM_ERR_HIP, "batch update+prepare kernel: LDS limit %zu: %s", lds, hipGetErrorString(e));
            return true;
        }
        have = lds;
    }
    double *wt_out = L.has_wt() ? reinterpret_cast<double *>(ws + L.off_wt) : nullptr;
    // workgroups: each redoes the update, the output (duplicates, fragments, bias, copies) is shared out by node
    // blocks -- 4 blocks of 16 nodes each at K = 400 (7 workgroups), 2 at K = 100 (4 workgroups)
    const int bper = xdim == 10 ? 2 : 4;
    const int grid = (L.nb + bper - 1) / bper;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kUpdThreads), lds, st, sa, c, reinterpret_cast<AssignHdr *>(ws),
                       reinterpret_cast<half8 *>(ws + L.off_wfrag), reinterpret_cast<f32x4 *>(ws + L.off_bias), wt_out, L.nb,
                       L.nch, L.cpl, L.idx_bits, pl, reinterpret_cast<float *>(ws + L.off_w32), L.cp32, L.npk);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) *rc = pxsom::hip_fail(e, "batch_update_prep_kernel");
    return true;
}

// Shapes the fused step kernel covers: the Pixie pixel SOM's -- 10 x 10 grid, even c <= 32, pair-aligned rows.
template <typename T>
bool step_fused_shape(const T *x, int64_t n, int c, int64_t ldx, int xdim, int ydim, int64_t group_stride)
{
    return xdim == kXD && ydim == kYD && n >= 1 && n < (int64_t)1 << 32 && c >= 2 && c <= 32 && c % 2 == 0 && ldx % 2 == 0 &&
           group_stride % 2 == 0 && reinterpret_cast<uintptr_t>(x) % (2 * sizeof(T)) == 0;
}

template <typename T>
int launch_batch_step(const T *x, int64_t n, int c, int64_t ldx, double *stats, const StepArgs &sa,
                      int tiles_per_wave, hipStream_t st)
{
    const int cpl = make_layout(n, c, kK).cpl;
    if (cpl == 6) return launch_step<T, 6>(x, n, c, ldx, stats, sa, tiles_per_wave, st);
    if (cpl == 8) return launch_step<T, 8>(x, n, c, ldx, stats, sa, tiles_per_wave, st);
    if (cpl == 4) return launch_step<T, 4>(x, n, c, ldx, stats, sa, tiles_per_wave, st);
    return launch_step<T, 2>(x, n, c, ldx, stats, sa, tiles_per_wave, st);
}

#define PXSOM_INSTANTIATE_STEP(T)                                                                                   \
    template bool step_fused_shape<T>(const T *, int64_t, int, int64_t, int, int, int64_t);                         \
    template int launch_batch_step<T>(const T *, int64_t, int, int64_t, double *, const StepArgs &, int, hipStream_t);
PXSOM_INSTANTIATE_STEP(float)
PXSOM_INSTANTIATE_STEP(double)
PXSOM_INSTANTIATE_STEP(_Float16)

#ifdef PXSOM_STEP_COUNT_LISTED
extern "C" __attribute__((visibility("default"))) int pxsom_dbg_listed(unsigned *out, int cap)
{
    static unsigned h[3][4096];
    unsigned nl = 0;
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(&nl, HIP_SYMBOL(g_dbg_launch), 4);
    (void)hipMemcpyFromSymbol(h[0], HIP_SYMBOL(g_dbg_listed), sizeof(h[0]));
    (void)hipMemcpyFromSymbol(h[1], HIP_SYMBOL(g_dbg_listed_max), sizeof(h[0]));
    (void)hipMemcpyFromSymbol(h[2], HIP_SYMBOL(g_dbg_rows), sizeof(h[0]));
    int m = 0;
    for (unsigned i = 1; i <= nl && i < 4096 && m + 3 <= cap; i++) {
        out[m++] = h[2][i];
        out[m++] = h[0][i];
        out[m++] = h[1][i];
    }
    return m / 3;
}
#endif
}  // namespace pxsom_bmu
