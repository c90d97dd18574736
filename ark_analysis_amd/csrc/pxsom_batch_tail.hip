// pxsom_batch_tail.hip -- the BMU-only tail of a batch training pass (K6b) as ONE persistent launch whose
// workgroups all sit on one XCD and talk through that XCD's L2 alone.
//
// Why: a tail step is 8 738 rows (config 2) and a chain of latencies -- as its own launch it costs ~12 us (launch gap,
// start-up spread, every workgroup redoing the update of all 100 nodes at two waves per SIMD of binary64 issue, statistics
// exchanged through device-scope atomics that live past the L2).  Measured on MI355X (scripts/ubench/xcd_local_sync.hip):
// a barrier among the 32 workgroups of one XCD made of one plain store + one polled line costs 0.40 us, against 3.7 us
// for a device-scope barrier over the chip.  So the tail runs where synchronisation is cheap:
//   * census (placement-independent; HIP promises nothing about where a workgroup lands): every workgroup of the launch
//     reads HW_REG_XCC_ID and takes a ticket; ticket 0 is the leader and its XCD the chosen one; workgroups elsewhere
//     leave at once; the others register and the leader closes the list when every ticket holder has decided (bounded
//     wait): P members (32 on MI355X), rank r.  No member ever waits for a workgroup that is not running;
//   * OWNER-COMPUTES update: member r owns the nodes r, r + P, ...  After a step's search every member writes its
//     binary64 table to its own slot (plain stores), flag barrier; the owner adds up the P slots of ITS nodes in slot
//     order (no atomics: the statistics are bit-reproducible), applies the pending update to them -- one gain per
//     node on one wave, not 100 of them on every CU -- converts them to the filter's binary16 fragments and publishes
//     node values, fragments, norms; flag barrier; everybody copies the 33 KB into LDS and searches its tiles;
//   * rows of the next round / the next step are requested right after the current ones are converted, a whole phase
//     before they are needed; plain stores + s_waitcnt vmcnt(0) + plain flag store publish, L1-bypassing (sc1) loads
//     consume: everything stays in the XCD's L2.
// Rule, arithmetic and state left behind (W of the last step in its wbuf slot, its statistics in its ring slot, the next
// ring slot cleared) are those of the launch-per-step route: orc_som_batch_sched (oracle/pxsom_oracle.c); reference call
// this stands in for: PixieSOMCluster.train_som, /root/reference/src/ark/phenotyping/cluster_helpers.py:98-116.
// Differences that cannot change a label: the filter's power-of-two scale is the one the PREVIOUS codebook would choose
// (any scale is valid; a codebook that outgrew binary16 under it sends every row to the exact path), and exact duplicates
// of an earlier node are not masked out (in a BMU-only step the first of two equal nodes takes all their rows and moves
// away; until then their rows are listed and settled exactly).
#include <algorithm>

#include "pxsom_batch_step.h"

namespace pxsom_bmu {
namespace {

#ifndef PXSOM_TAIL_WAVES
#define PXSOM_TAIL_WAVES 8
#endif
#ifndef PXSOM_TAIL_TPR
#define PXSOM_TAIL_TPR 3
#endif
constexpr int kTailWaves = PXSOM_TAIL_WAVES, kTailThreads = 64 * kTailWaves, kTPR = PXSOM_TAIL_TPR;   // tiles of 16 rows per wave and round
constexpr int kMaxMembers = 64;
constexpr unsigned kClosed = 0x80000000u;

#ifdef PXSOM_TAIL_TIMING
// scripts/ubench/tail_phase_timing.hip: s_memrealtime (100 MHz) stamps of every member's thread 0 inside one chosen step
__device__ long long g_tail_ticks[kMaxMembers][16];
__device__ int g_tail_stamp_step = 8;
#define TAIL_STAMP(i)                                                                         \
    do {                                                                                      \
        if (threadIdx.x == 0 && s == g_tail_stamp_step) g_tail_ticks[rank][i] = (long long)wall_clock64(); \
    } while (0)
#else
#define TAIL_STAMP(i) \
    do {              \
    } while (0)
#endif

struct TailCtl {
    unsigned tickets, chosen, members, decided, p_pub, pad[11];
    unsigned flags_a[kMaxMembers];   // "my table is in my slot" (value: steps done)
    unsigned flags_b[kMaxMembers];   // "my nodes are published"
};

// What the owners publish per step, in the order (and byte layout) the members keep it in LDS -- a member's copy is one
// flat sweep of 16-byte loads: codebook transposed [c][K] | fragments [NB][hi, lo][64] half8 | bias [NB][64] f32x4 |
// squared norms [K] | largest magnitudes [K] (of the centred nodes)
struct PubLayout {
    size_t wt, frag, bias, nrm, mx, total;
};
__host__ __device__ inline PubLayout pub_layout(int c)
{
    PubLayout p;
    size_t o = 0;
    p.wt = o;    o += (size_t)c * kK * 8;
    p.frag = o;  o += (size_t)kNB * 2 * 64 * 16;
    p.bias = o;  o += (size_t)kNB * 64 * 16;
    p.nrm = o;   o += (size_t)kK * 8;
    p.mx = o;    o += (size_t)kK * 8;
    p.total = o;   // a multiple of 16 (c even, K = 100)
    return p;
}

// scratch in HBM (caller's workspace): control words | member tables | published codebook
struct TailScratch {
    size_t ctl, slots, pub, total;
};
__host__ __device__ inline TailScratch tail_scratch(int c)
{
    TailScratch s;
    size_t o = 0;
    const size_t nstats = (size_t)kK * c + kK;
    s.ctl = o;       o += 1024;
    s.slots = o;     o += (size_t)kMaxMembers * nstats * 8;
    s.pub = o;       o += pub_layout(c).total;
    s.total = (o + 255) / 256 * 256;
    return s;
}
static_assert(sizeof(TailCtl) <= 1024, "control block");

struct TailLds {
    size_t ls, pub, tl, wown, ovf, hdr, mu, total;
};
__host__ __device__ inline TailLds tail_lds(int c)
{
    TailLds L;
    size_t o = 0;
    L.ls = o;    o += ((size_t)kK * c + kK) * 8;          // table [K*c sums | K counts]
    L.pub = o;   o += pub_layout(c).total;                // this member's copy of what the owners published
    L.tl = o;    o += (size_t)kK * (c + 1) * 8;           // owner: summed statistics of its nodes; search: queue of listed rows
    L.wown = o;  o += (size_t)kK * c * 8;                 // owner: current values of its nodes [own node][c]
    L.ovf = o;   o += (size_t)kTailWaves * 32 * 8;        // one row per wave (queue overflow)
    L.hdr = o;   o += 64;
    L.mu = o;    o += 40 * 4;
    L.total = o;
    return L;
}

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xfu; }   // HW_REG_XCC_ID[3:0]
// L1-bypassing loads (sc1): served by the XCD's L2, which is where the other members' stores are
__device__ __forceinline__ unsigned ld_l2(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_l2(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// power-of-two scale with maxabs * scale in [128, 256), capped by what the centring vector's own norm would choose + 2^6
// (batch_step_kernel P4)
__device__ __forceinline__ double scale_for(double maxabs, float mu_norm)
{
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
    return ldexp(1.0, e);
}

// One word of the statistics summed over the members' slots by 4 adjacent lanes (lane part h takes the slots h, h + 4, ...):
// eight loads per lane in flight at once, no branch between them (a slot index past the end re-reads the last slot and
// adds +0.0), then a butterfly over the quad.  The order is fixed, so the sums are reproducible.
__device__ __forceinline__ double sum_slots(const double *src, int nslots, int nstats, int h)
{
    double acc = 0.0;
    for (int p0 = 0; p0 < nslots; p0 += 32) {
        double v[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int p = p0 + h + 4 * r;
#if defined(PXSOM_TAIL_SLOT_MODE) && PXSOM_TAIL_SLOT_MODE == 1      // timing experiment: plain loads (stale data possible)
            v[r] = *(const volatile double *)(src + (size_t)(p < nslots ? p : nslots - 1) * nstats);
#elif defined(PXSOM_TAIL_SLOT_MODE) && PXSOM_TAIL_SLOT_MODE == 2    // timing experiment: every load from slot 0
            v[r] = ld_l2(src + (size_t)(p < 0 ? p : 0) * nstats);
#elif defined(PXSOM_TAIL_SLOT_MODE) && PXSOM_TAIL_SLOT_MODE == 4    // timing experiment: nt loads
            v[r] = __builtin_nontemporal_load(src + (size_t)(p < nslots ? p : nslots - 1) * nstats);
#else
            v[r] = ld_l2(src + (size_t)(p < nslots ? p : nslots - 1) * nstats);
#endif
        }
#pragma unroll
        for (int r = 0; r < 8; r++) acc += (p0 + h + 4 * r < nslots) ? v[r] : 0.0;
    }
    acc += pxsom::dpp_f64(acc, 0);   // quad_perm [1,0,3,2]
    acc += pxsom::dpp_f64(acc, 1);   // quad_perm [2,3,0,1]
    return acc;
}

// sum / maximum / or over the 32 lanes of a half wave, result in all of them: DPP inside the rows of 16, one
// v_permlane16_swap across them (no LDS round trips)
struct D2 {
    double a, b;
};
// {own, partner} of lane l and lane l ^ 16, in lane-dependent order: fed to symmetric functions only
__device__ __forceinline__ D2 xchg16_f64(double v)
{
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const unsigned lo = (unsigned)__double_as_longlong(v), hi = (unsigned)(__double_as_longlong(v) >> 32);
    const u2 rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false), rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return {__longlong_as_double(((long long)rh[0] << 32) | rl[0]), __longlong_as_double(((long long)rh[1] << 32) | rl[1])};
}
__device__ __forceinline__ double half_wave_sum(double v)
{
#pragma unroll
    for (int st = 0; st < 4; st++) v += pxsom::dpp_f64(v, st);
    const D2 r = xchg16_f64(v);
    return r.a + r.b;
}
__device__ __forceinline__ double half_wave_max(double v)
{
#pragma unroll
    for (int st = 0; st < 4; st++) v = fmax(v, pxsom::dpp_f64(v, st));
    const D2 r = xchg16_f64(v);
    return fmax(r.a, r.b);
}

template <typename T, int CPL>
__global__ __launch_bounds__(kTailThreads) void batch_tail_kernel(const T *__restrict__ x, int c, int64_t ldx, TailArgs ta)
{
    extern __shared__ __attribute__((aligned(16))) char tail_smem[];
    const TailLds L = tail_lds(c);
    const PubLayout PL = pub_layout(c);
    double *ls = reinterpret_cast<double *>(tail_smem + L.ls);
    char *pub_l = tail_smem + L.pub;                       // this member's copy of the published codebook
    double *wt = reinterpret_cast<double *>(pub_l + PL.wt);
    half8 *frag_l = reinterpret_cast<half8 *>(pub_l + PL.frag);
    f32x4 *bias_l = reinterpret_cast<f32x4 *>(pub_l + PL.bias);
    double *nrm_l = reinterpret_cast<double *>(pub_l + PL.nrm);
    double *mx_l = reinterpret_cast<double *>(pub_l + PL.mx);
    double *tl = reinterpret_cast<double *>(tail_smem + L.tl);
    double *wown = reinterpret_cast<double *>(tail_smem + L.wown);
    double *ovf = reinterpret_cast<double *>(tail_smem + L.ovf);
    StepHdr *hdr = reinterpret_cast<StepHdr *>(tail_smem + L.hdr);
    float *mu_l = reinterpret_cast<float *>(tail_smem + L.mu);
    __shared__ unsigned s_rank, s_p;
    __shared__ double s_red[kTailWaves];

    const TailScratch SC = tail_scratch(c);
    TailCtl *ctl = reinterpret_cast<TailCtl *>(ta.scratch + SC.ctl);
    double *slots = reinterpret_cast<double *>(ta.scratch + SC.slots);
    char *pub_g = ta.scratch + SC.pub;
    double *pub_wt = reinterpret_cast<double *>(pub_g + PL.wt);
    _Float16 *pub_frag = reinterpret_cast<_Float16 *>(pub_g + PL.frag);
    float *pub_bias = reinterpret_cast<float *>(pub_g + PL.bias);
    double *pub_nrm = reinterpret_cast<double *>(pub_g + PL.nrm);
    double *pub_mx = reinterpret_cast<double *>(pub_g + PL.mx);

    constexpr int NP = CPL / 2;
    typedef typename Pair<T>::type P2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pix = lane & 15, q = lane >> 4;
    const int nstats = kK * c + kK, NC = c + 1;

    // ---- census: who is a member ---------------------------------------------------------------------------------
    if (tid == 0) {
        const unsigned xcc = xcc_id();
        unsigned rank = 0xffffffffu, p = 0;
        const unsigned t = __hip_atomic_fetch_add(&ctl->tickets, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == 0) {
            __hip_atomic_store(&ctl->chosen, xcc + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            rank = __hip_atomic_fetch_add(&ctl->members, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // every other workgroup of the launch reports once it has registered or decided not to (bounded: a launch whose
            // workgroups are not all resident -- the GPU shared with another process -- goes on with those that are)
            int spins = 0;
            while (__hip_atomic_load(&ctl->decided, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x - 1u && spins++ < (1 << 15))
                __builtin_amdgcn_s_sleep(1);
            const unsigned cnt = __hip_atomic_fetch_or(&ctl->members, kClosed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            p = cnt < (unsigned)kMaxMembers ? cnt : (unsigned)kMaxMembers;
            __hip_atomic_store(&ctl->p_pub, p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            unsigned ch = 0;
            int spins = 0;
            while ((ch = __hip_atomic_load(&ctl->chosen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u && spins++ < (1 << 22))
                __builtin_amdgcn_s_sleep(1);
            if (ch == xcc + 1u) {
                const unsigned old = __hip_atomic_fetch_add(&ctl->members, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(&ctl->decided, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (!(old & kClosed) && old < (unsigned)kMaxMembers) {
                    rank = old;
                    while ((p = __hip_atomic_load(&ctl->p_pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u) __builtin_amdgcn_s_sleep(1);
                }
            } else {
                __hip_atomic_fetch_add(&ctl->decided, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        s_rank = rank;
        s_p = p;
    }
    __syncthreads();
    if (s_rank >= s_p) return;   // not a member (0xffffffff), or registered past the list's end
    const int rank = (int)s_rank, P = (int)s_p;
    const int nown = rank < kK ? (kK - rank + P - 1) / P : 0;       // this member's nodes: rank, rank + P, ...
    // tile T of a step belongs to member T % P, there to wave (T / P) % waves: every member gets its share to within one
    // tile whatever the step's size.  Tile j of this wave: (j * waves + wv) * P + rank
    const int Wt = P * kTailWaves, u = wv * P + rank;
    const double qmagic = sizeof(T) == 8 ? ta.qmagic : 0.0;

    // sense-free flag barrier: a plain store of the step count into the member's word, the first wave polls the P words
    auto flag_barrier = [&](unsigned *flags, unsigned val) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's payload stores have reached the L2
        __syncthreads();
        if (tid == 0) {
            flags[rank] = val;
        }
        if (wv == 0) {
            for (;;) {
                const unsigned f = lane < P ? ld_l2(flags + lane) : val;
                if (__ballot(f < val) == 0ull) break;
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
    };

    // ---- rows: requested a whole phase before they are needed -----------------------------------------------------
    P2 raw[kTPR][NP];
    auto load_round = [&](const TailStep &S, int j0) {
#pragma unroll
        for (int t = 0; t < kTPR; t++) {
            long long row = ((long long)(j0 + t) * Wt + u) * 16 + pix;
            if (row > S.rows - 1) row = S.rows - 1;
            const unsigned w = (unsigned)S.width;
            const unsigned grp = (unsigned)row / w, sub = (unsigned)row - grp * w;
            const T *rp = x + ((int64_t)grp * ta.phases + S.e0 + sub) * ldx;
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
    load_round(ta.st[0], 0);

    // ---- one-off set-up --------------------------------------------------------------------------------------------
    if (tid < 40) mu_l[tid] = (tid < 33 && ta.mu32) ? ta.mu32[tid < 32 ? tid : kFilterMaxChannels] : 0.f;   // 32 channels + the norm
    for (int e = tid; e < nstats; e += kTailThreads) ls[e] = 0.0;
    if (tid == 0) {
        hdr->q_n = 0u;
        hdr->bad = 0;
    }
    // this member's nodes: current values into LDS; their fragment elements start from zero (the channel slots past c and
    // past CPL are never written again)
    for (int e = tid; e < nown * c; e += kTailThreads) {
        const int i = e / c, j = e - i * c;
        wown[e] = ta.w_in[(size_t)(rank + i * P) * c + j];
    }
    for (int e = tid; e < nown * 8 * 8; e += kTailThreads) {   // 8 half8 per node (4 lane groups x hi / lo) = 64 halves
        const int i = e >> 6, r = e & 63, k = rank + i * P;
        const int b = k < 96 ? k >> 4 : 6, m = k < 96 ? k & 15 : 4 * (k - 96);
        const int h = r >> 5, nq = (r >> 3) & 3, ii = r & 7;
        pub_frag[((size_t)(b * 2 + h) * 64 + ((nq << 4) | m)) * 8 + ii] = (_Float16)0;
    }
    if (rank == 0 && tid < 48) {   // the 12 rows of the last node block that hold no node: zero fragments, bias -inf, for good
        const int i = tid >> 2, nq = tid & 3, m = (i / 3) * 4 + (i % 3) + 1;
        for (int ii = 0; ii < 8; ii++) {
            pub_frag[((size_t)(6 * 2 + 0) * 64 + ((nq << 4) | m)) * 8 + ii] = (_Float16)0;
            pub_frag[((size_t)(6 * 2 + 1) * 64 + ((nq << 4) | m)) * 8 + ii] = (_Float16)0;
        }
        for (int u4 = 0; u4 < 4; u4++) pub_bias[((size_t)(6 * 64 + (m >> 2) * 16 + nq * 4 + u4)) * 4 + (m & 3)] = kNegBig;
    }
    __syncthreads();
    const float mu_norm = mu_l[32];
    double scale;   // the power-of-two scale of the step's fragments: what the codebook BEFORE the update would choose
    {
        double mymax = 0.0;
        for (int e = tid; e < kK * c; e += kTailThreads) {
            const int ch = e % c;
            mymax = fmax(mymax, fabs(ta.w_in[e] - (double)mu_l[ch < 32 ? ch : 0]));
        }
        const double wmax = -pxsom::wave_min_f64(-((mymax == mymax) ? mymax : 0.0));
        if (lane == 0) s_red[wv] = wmax;
        __syncthreads();
        double maxabs = s_red[0];
#pragma unroll
        for (int i = 1; i < kTailWaves; i++) maxabs = fmax(maxabs, s_red[i]);
        scale = scale_for(maxabs, mu_norm);
        __syncthreads();
    }

    const float tol_rel = ta.tol_rel, tol_abs = ta.tol_abs, x_limit = 60000.0f;
    double *qrows = tl;
    constexpr unsigned idx_mask = 127u;
    // the published codebook as one buffer of 16-byte quads, read past the L1 (sc1)
    const __amdgpu_buffer_rsrc_t pub_rsrc = __builtin_amdgcn_make_buffer_rsrc(pub_g, 0, (int)PL.total, 0x00020000);
    const int pub_quads = (int)(PL.total / 16);
    typedef unsigned uint4v __attribute__((ext_vector_type(4)));

    for (int s = 0; s < ta.nsteps; s++) {
        const TailStep S = ta.st[s];
        // ---- A: the owner adds up its nodes' statistics (4 lanes per word, the slots dealt round robin, every load in
        // flight at once) and applies the pending update ---------------------------------------------------------------
        const bool has_upd = s > 0 || ta.first_has_update != 0;
        TAIL_STAMP(0);
        if (has_upd) {
            const double *base = s > 0 ? slots : ta.stats_first;
            const int nslots = s > 0 ? P : 1;
            for (int e0 = 0; e0 < nown * NC; e0 += kTailThreads / 4) {
                const int e = e0 + (tid >> 2), h = tid & 3;
                const bool act = e < nown * NC;
                const int ec = act ? e : 0;
                const int i = ec / NC, j = ec - i * NC, k = rank + i * P;
                const double *src = base + (j < c ? (size_t)k * c + j : (size_t)kK * c + k);
                const double acc = sum_slots(src, nslots, nstats, h);
                if (act && h == 0) tl[e] = acc;
            }
        }
        __syncthreads();
        TAIL_STAMP(1);
        // update + publish: thread <-> (own node i, channel j), 32 lanes per node: norm and maximum by butterflies
        for (int i0 = 0; i0 < nown; i0 += kTailThreads / 32) {
            const int i = i0 + (tid >> 5), j = tid & 31;
            const bool act = i < nown && j < c;
            const int ic = i < nown ? i : 0, jc = j < c ? j : 0, k = rank + ic * P;
            double v = wown[ic * c + jc];
            {
#pragma clang fp contract(off)
                if (has_upd) {
                    const double den = 0.0 + tl[ic * NC + c];
                    if (den > 0.0) {
                        // gain = 1 - (1-alpha)^den (batch_gain; orc_batch_update)
                        const double gain = batch_gain(den, S.q, S.sat), inv = 1.0 / den;
                        const double num = 0.0 + tl[ic * NC + jc];
                        v = gain == 1.0 ? num * inv : v + gain * (num * inv - v);
                    }
                }
            }
            const double vc = v - (double)mu_l[jc];
            double nrm = act ? vc * vc : 0.0, mymax = act ? fabs(vc) : 0.0;
            // a non-finite value poisons the node's maximum: one reduction serves both
            if (act && !(fabs(v) <= DBL_MAX)) mymax = __builtin_inf();
            nrm = half_wave_sum(nrm);
            mymax = half_wave_max(mymax);
            const bool bad = !(mymax <= DBL_MAX);
            const int b = k < 96 ? k >> 4 : 6, m = k < 96 ? k & 15 : 4 * (k - 96);
            if (act) {
                wown[ic * c + jc] = v;
                pub_wt[(size_t)jc * kK + k] = v;
                const float W = (float)(vc * scale);
                const _Float16 hi = (_Float16)W, lo = (_Float16)(W - (float)hi);
                const int nq = jc / CPL, ii = jc - nq * CPL, lf = (nq << 4) | m;
                pub_frag[((size_t)(b * 2 + 0) * 64 + lf) * 8 + ii] = hi;
                pub_frag[((size_t)(b * 2 + 1) * 64 + lf) * 8 + ii] = lo;
            }
            if (i < nown && j < 16)   // bias of accumulator row m, replicated over the 16 pixel lanes
                pub_bias[((size_t)(b * 64 + (m >> 2) * 16 + j)) * 4 + (m & 3)] = bad ? kNegBig : (float)(-0.5 * nrm * scale * scale);
            if (i < nown && j == 0) {
                pub_nrm[k] = bad ? __builtin_nan("") : nrm;
                pub_mx[k] = bad ? __builtin_inf() : mymax;
            }
        }
        // ---- B: everybody's nodes are out ---------------------------------------------------------------------------
        TAIL_STAMP(2);
        TAIL_STAMP(3);
        flag_barrier(ctl->flags_b, (unsigned)(s + 1));
        TAIL_STAMP(4);
        // ---- C: the new codebook into LDS: one flat sweep, every load in flight before the first LDS write ------------
        {
            constexpr int kQ = 6;   // 16-byte quads per thread (c <= 32: 3 044 quads)
            uint4v qv[kQ];
#pragma unroll
            for (int r = 0; r < kQ; r++) {
                const int qi = tid + r * kTailThreads;
                qv[r] = __builtin_amdgcn_raw_buffer_load_b128(pub_rsrc, (qi < pub_quads ? qi : pub_quads - 1) * 16, 0, 1 << 4 /* sc1 */);
            }
#pragma unroll
            for (int r = 0; r < kQ; r++) {
                const int qi = tid + r * kTailThreads;
                if (qi < pub_quads) reinterpret_cast<uint4v *>(pub_l)[qi] = qv[r];
            }
        }
        __syncthreads();
        TAIL_STAMP(5);
        float fscale, wn_max, abs_up;
        bool force_exact;
        double scale_next;
        {
            // every wave reduces the 100 norms / magnitudes for itself (no further barrier)
            const double n0 = nrm_l[lane], n1 = nrm_l[lane + 64 < kK ? lane + 64 : lane];
            const double x0 = mx_l[lane], x1 = mx_l[lane + 64 < kK ? lane + 64 : lane];
            const bool badn = !(n0 == n0) || !(n1 == n1) || !(x0 <= DBL_MAX) || !(x1 <= DBL_MAX);
            const bool anybad = __ballot(badn) != 0ull;
            const double maxabs = -pxsom::wave_min_f64(-(badn ? 0.0 : fmax(x0, x1)));
            const double wn2max = -pxsom::wave_min_f64(-(badn ? 0.0 : fmax(n0, n1)));
            // the fragments were formed with the previous codebook's scale: fine while the new one stays inside binary16
            const bool badw = anybad || !(wn2max * scale * scale <= 1.0e30) || !(maxabs * scale < 32768.0);
            fscale = (float)scale;
            wn_max = badw ? 0.f : (float)(sqrt(wn2max) * scale * (1.0 + 1e-6));
            force_exact = badw;
            // tol_abs holds the cuts inside an MFMA's groups for a scaled codebook below 256 (filter_cut_abs); this step's fragments
            // carry the previous codebook's scale, so the largest magnitude may have left that range: scale the term with it
            abs_up = fmaxf(1.f, (float)(maxabs * scale * (1.0 / 256.0) * (1.0 + 1e-6)));
            scale_next = scale_for(maxabs, mu_norm);
        }
        TAIL_STAMP(6);

        // ---- D: BMU search of this wave's tiles ---------------------------------------------------------------------
        float mus[NP][2];
#pragma unroll
        for (int p = 0; p < NP; p++) {
            int ch = q * CPL + 2 * p;
            if (ch > c - 2) ch = c - 2;
            mus[p][0] = mu_l[ch] * fscale;
            mus[p][1] = mu_l[ch + 1] * fscale;
        }
        const long long ntiles = (S.rows + 15) >> 4;
        const int my_tiles = ntiles > u ? (int)((ntiles - u + Wt - 1) / Wt) : 0;
        bool next_requested = false;
        for (int j0 = 0; j0 < my_tiles; j0 += kTPR) {
            half8 bh[kTPR], bl[kTPR];
            float ss[kTPR];
            P2 cur[kTPR][NP];
#pragma unroll
            for (int t = 0; t < kTPR; t++) {
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
#pragma unroll
                for (int p = 0; p < NP; p++) cur[t][p] = raw[t][p];
            }
            if (j0 == 0) TAIL_STAMP(12);
            // the next round's rows -- or the next step's first round
            if (j0 + kTPR < my_tiles) {
                load_round(S, j0 + kTPR);
            } else if (s + 1 < ta.nsteps) {
                load_round(ta.st[s + 1], 0);
                next_requested = true;
            }
            float m1[kTPR], m2[kTPR];
#pragma unroll
            for (int t = 0; t < kTPR; t++) m1[t] = m2[t] = kNegBig;
#pragma unroll
            for (int b = 0; b < kNB; b++) {
                const half8 wa0 = frag_l[(b * 2 + 0) * 64 + lane], wa1 = frag_l[(b * 2 + 1) * 64 + lane];
                const f32x4 bb = bias_l[b * 64 + lane];
                f32x4 acc[kTPR];
#pragma unroll
                for (int t = 0; t < kTPR; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa0, bh[t], bb, 0, 0, 0);
#pragma unroll
                for (int t = 0; t < kTPR; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa0, bl[t], acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < kTPR; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa1, bh[t], acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < kTPR; t++) {
                    if (b < kNB - 1) {
                        top2_quad(m1[t], m2[t], pack_idx(acc[t][0], (unsigned)(b * 4 + 0), idx_mask),
                                  pack_idx(acc[t][1], (unsigned)(b * 4 + 1), idx_mask),
                                  pack_idx(acc[t][2], (unsigned)(b * 4 + 2), idx_mask),
                                  pack_idx(acc[t][3], (unsigned)(b * 4 + 3), idx_mask));
                    } else {   // last block: only accumulator register 0 holds real nodes (K = 100)
                        const float p0 = pack_idx(acc[t][0], (unsigned)(b * 4 + 0), idx_mask);
                        m2[t] = __builtin_amdgcn_fmed3f(m1[t], m2[t], p0);
                        m1[t] = fmaxf(m1[t], p0);
                    }
                }
            }
            if (j0 == 0) TAIL_STAMP(13);
#pragma unroll
            for (int t = 0; t < kTPR; t++) {
                // merge of the 4 lane groups that share a pixel: afterwards all four hold the pixel's top-2
                float a1 = __uint_as_float(__float_as_uint(m1[t]) | ((unsigned)q << 5)), a2 = m2[t], s2 = ss[t];
                {
                    const F2 e1 = xchg16(a1), e2 = xchg16(a2), es = xchg16(s2);
                    a1 = fmaxf(e1.a, e1.b);
                    a2 = fmaxf(fmaxf(fminf(e1.a, e1.b), e2.a), e2.b);
                    s2 = es.a + es.b;
                }
                {
                    const F2 e1 = xchg32(a1), e2 = xchg32(a2), es = xchg32(s2);
                    a1 = fmaxf(e1.a, e1.b);
                    a2 = fmaxf(fmaxf(fminf(e1.a, e1.b), e2.a), e2.b);
                    s2 = es.a + es.b;
                }
                const float xn = __builtin_amdgcn_sqrtf(s2) * 1.001f;
                const float tol = tol_rel * (xn * wn_max + 0.5f * wn_max * wn_max) + tol_abs * abs_up * (xn + wn_max);
                const unsigned nonfinite = (unsigned)((__float_as_uint(s2) & 0x7f800000u) == 0x7f800000u);
                const long long row = ((long long)(j0 + t) * Wt + u) * 16 + pix;
                const bool valid = row < S.rows;
                const bool amb = valid && ((!((a1 - a2) > tol)) || !(xn < x_limit) || nonfinite != 0u || force_exact);
                const unsigned id = __float_as_uint(a1) & idx_mask;
                const unsigned wq = id >> 5, wb = (id >> 2) & 7u, wr = id & 3u;
                const unsigned real = wb == (unsigned)(kNB - 1) ? 16u * wb + 4u * wr + wq : 16u * wb + 4u * wq + wr;
                if (valid && !amb) {
                    double *dst = ls + (size_t)real * c + q * CPL;
#pragma unroll
                    for (int p = 0; p < NP; p++) {
                        if (q * CPL + 2 * p <= c - 2) {   // clamped slots re-read the last pair: not theirs
                            double v0 = (double)cur[t][p].x, v1 = (double)cur[t][p].y;
                            if constexpr (sizeof(T) == 8) {
                                v0 = qround(v0, qmagic);
                                v1 = qround(v1, qmagic);
                            }
                            __hip_atomic_fetch_add(dst + 2 * p, v0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            __hip_atomic_fetch_add(dst + 2 * p + 1, v1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
                    if (q == 0) __hip_atomic_fetch_add(ls + (size_t)kK * c + real, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
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
                        exact_row_from_lds(slot, c, wt, ls, lane, qmagic);
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            }
        }
        TAIL_STAMP(14);
        if (!next_requested && s + 1 < ta.nsteps) load_round(ta.st[s + 1], 0);
        // ---- E: listed rows, by whichever wave is free ---------------------------------------------------------------
        TAIL_STAMP(7);
        __syncthreads();
        TAIL_STAMP(8);
        {
            const unsigned queued = hdr->q_n < (unsigned)kQueueRows ? hdr->q_n : (unsigned)kQueueRows;
            for (unsigned i = wv; i < queued; i += kTailWaves) exact_row_from_lds(qrows + (size_t)i * c, c, wt, ls, lane, qmagic);
        }
        __syncthreads();
        TAIL_STAMP(9);
        // ---- F: the table into this member's slot; cleared for the next step ----------------------------------------
        {
            double *mine = slots + (size_t)rank * nstats;
            for (int e = tid; e < nstats; e += kTailThreads) {
                mine[e] = ls[e];
                ls[e] = 0.0;
            }
            if (tid == 0) {
                hdr->q_n = 0u;
                hdr->bad = 0;
            }
        }
        scale = scale_next;
        TAIL_STAMP(10);
        flag_barrier(ctl->flags_a, (unsigned)(s + 1));
        TAIL_STAMP(11);
    }

    // ---- what the per-step route leaves behind: the last step's statistics and codebook, the next ring slot cleared;
    // optionally the run's last update as well
    for (int e0 = 0; e0 < nown * NC; e0 += kTailThreads / 4) {
        const int e = e0 + (tid >> 2), h = tid & 3;
        const bool act = e < nown * NC;
        const int ec = act ? e : 0;
        const int i = ec / NC, j = ec - i * NC, k = rank + i * P;
        const size_t idx = j < c ? (size_t)k * c + j : (size_t)kK * c + k;
        const double acc = sum_slots(slots + idx, P, nstats, h);
        if (act && h == 0) {
            tl[e] = acc;
            ta.stats_last[idx] = acc;
        }
    }
    __syncthreads();
    {
#pragma clang fp contract(off)
        for (int e = tid; e < nown * c; e += kTailThreads) {
            const int i = e / c, j = e - i * c, k = rank + i * P;
            double v = wown[e];
            ta.w_last[(size_t)k * c + j] = v;
            if (ta.final_update) {
                const double den = 0.0 + tl[i * NC + c];
                if (den > 0.0) {
                    const double gain = batch_gain(den, ta.q_final, ta.sat_final), inv = 1.0 / den;
                    const double num = 0.0 + tl[i * NC + j];
                    v = gain == 1.0 ? num * inv : v + gain * (num * inv - v);
                }
                ta.w_final[(size_t)k * c + j] = v;
            }
        }
    }
    if (ta.stats_zero) {
        const int per = (nstats + P - 1) / P;
        const int e1 = min((rank + 1) * per, nstats);
        for (int e = rank * per + tid; e < e1; e += kTailThreads) ta.stats_zero[e] = 0.0;
    }
}

}  // namespace

size_t tail_scratch_bytes(int c) { return tail_scratch(c).total; }

template <typename T, int CPL>
int launch_tail(const T *x, int c, int64_t ldx, const TailArgs &ta, hipStream_t st)
{
    // at least 81 KB: one workgroup per CU, so that the members of the chosen XCD sit on distinct CUs
    const size_t lds = std::max<size_t>(tail_lds(c).total, 81 * 1024);
    auto kern = batch_tail_kernel<T, CPL>;
    static pxsom::PerDevice<size_t> attr_lds_on;
    size_t &attr_lds = attr_lds_on.here();
    if (attr_lds < lds) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess)
            return pxsom::fail(PXSOM_ERR_HIP, "batch tail kernel: cannot raise the LDS limit to %zu bytes: %s", lds, hipGetErrorString(e));
        attr_lds = lds;
    }
    PXSOM_HIP_TRY(hipMemsetAsync(ta.scratch + tail_scratch(c).ctl, 0, sizeof(TailCtl), st));
    const int grid = pxsom::device_cu_count();
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kTailThreads), lds, st, x, c, ldx, ta);
    PXSOM_LAUNCH_CHECK("batch_tail_kernel");
    return PXSOM_OK;
}

template <typename T>
int launch_batch_tail(const T *x, int c, int64_t ldx, const TailArgs &ta, hipStream_t st)
{
    const int cpl = make_layout(1, c, kK).cpl;
    if (cpl == 6) return launch_tail<T, 6>(x, c, ldx, ta, st);
    if (cpl == 8) return launch_tail<T, 8>(x, c, ldx, ta, st);
    if (cpl == 4) return launch_tail<T, 4>(x, c, ldx, ta, st);
    return launch_tail<T, 2>(x, c, ldx, ta, st);
}

template int launch_batch_tail<float>(const float *, int, int64_t, const TailArgs &, hipStream_t);
template int launch_batch_tail<double>(const double *, int, int64_t, const TailArgs &, hipStream_t);
template int launch_batch_tail<_Float16>(const _Float16 *, int, int64_t, const TailArgs &, hipStream_t);

}  // namespace pxsom_bmu
