// pxsom_sums.hip -- per-cluster sums with wave-private LDS tables, two channels per lane.
//
// Same scheme as cluster_sums_private_kernel (pxsom_train.hip): every wave owns a [k + 1, c] binary64 table
// and updates it with plain read / add / write, the lanes of one instruction touching distinct words unless
// two of its rows carry the same label (then the group is applied row by row).  Here a lane holds a channel
// PAIR: c / 2 lanes per row, RPI = 64 / (c / 2) rows per instruction (5 at c = 22 instead of 2), one 8-byte
// load and one ds_read_b128 / ds_write_b128 per lane and group.  That kernel was bound by its instruction
// count (one dword per lane: the load stream alone ran at 4.6 TB/s); this one issues 2.5x fewer per row.
// With more rows per instruction a repeat inside a group is likelier (10 % at K = 100, RPI = 5), so groups
// are not paired up: one group = one unit.
#include <algorithm>

#include "pxsom_common.h"
#include "pxsom_sums.h"

namespace pxsom {
namespace {

typedef double d2 __attribute__((ext_vector_type(2)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));

template <typename T>
struct PairBits;
template <>
struct PairBits<float> {
    typedef u2 type;
    static __device__ __forceinline__ type load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
    {
        return __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    }
    static __device__ __forceinline__ d2 widen(type v)
    {
        return d2{(double)__uint_as_float(v[0]), (double)__uint_as_float(v[1])};
    }
};
template <>
struct PairBits<_Float16> {
    typedef unsigned type;
    static __device__ __forceinline__ type load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
    {
        return __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0);
    }
    static __device__ __forceinline__ d2 widen(type v)
    {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const h2 h = __builtin_bit_cast(h2, v);
        return d2{(double)h[0], (double)h[1]};
    }
};

template <>
struct PairBits<double> {
    typedef unsigned type __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ type load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
    {
        return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    }
    static __device__ __forceinline__ d2 widen(type v) { return __builtin_bit_cast(d2, v); }
};

template <typename T, int RPI, bool COUNT_F64>
__global__ __launch_bounds__(256) void cluster_sums_pairs_kernel(const T *__restrict__ x, int64_t n, int c,
                                                                 int64_t ldx, const int32_t *__restrict__ labels,
                                                                 int k, double *sums, unsigned long long *counts,
                                                                 int64_t rows_per_wave)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    typedef typename PairBits<T>::type bits_t;
    constexpr int U = sizeof(T) == 8 ? 16 : 32;   // groups per tile == loads in flight per lane
    constexpr int TR = RPI * U;             // rows per tile
    constexpr int RL = 64 / RPI * RPI;      // rows per label register (whole groups)
    constexpr int NL = (TR + RL - 1) / RL;  // label registers per tile
    const int tid = threadIdx.x, bd = blockDim.x, lane = tid & 63, nwv = bd >> 6;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tstride = (k + 1) * c + 128;  // doubles per wave table (+ a spare pair per lane)
    double *all = reinterpret_cast<double *>(smem_raw);
    double *tbl = all + (size_t)wv * tstride;
    unsigned *cnt = reinterpret_cast<unsigned *>(all + (size_t)nwv * tstride);  // [k], shared by the waves
    for (int e = tid; e < nwv * tstride; e += bd) all[e] = 0.0;
    for (int e = tid; e < k; e += bd) cnt[e] = 0u;
    __syncthreads();

    const int pairs = c >> 1;
    const int slot = lane / pairs, pr = lane - slot * pairs;
    const bool active = slot < RPI;
    const unsigned lane_off = active ? (unsigned)((slot * ldx + 2 * pr) * (int64_t)sizeof(T)) : 0u;  // bytes
    const int64_t gw = (int64_t)blockIdx.x * nwv + wv;
    const int64_t ra = gw * rows_per_wave;
    const int64_t rb = ra + rows_per_wave < n ? ra + rows_per_wave : n;
    if (ra < rb) {  // wave-uniform
        // whole groups end at row `lim` (relative to ra); the rows behind it (last wave only) are added one by
        // one.  Every load is unconditional with a clamped, wave-uniform row (see cluster_sums_private_kernel).
        const int span = (int)(rb - ra), lim = span - span % RPI;
        const int last = (int)(n - 1 - ra);
        const unsigned gstep = (unsigned)(RPI * ldx * (int64_t)sizeof(T));
        const unsigned safe = (unsigned)((ra + RPI <= n ? 0 : n - RPI - ra) * ldx * (int64_t)sizeof(T));
        const __amdgpu_buffer_rsrc_t xres =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(x + ra * ldx), 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t lres =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(labels + ra), 0, 0x7fffffff, 0x00020000);
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
        auto load_val = [&](int rel, unsigned off) -> bits_t {
            return PairBits<T>::load(xres, lane_off, rel + RPI <= lim ? off : safe);
        };
        // a label repeated inside a group: that group is applied row by row.  (Rounds by occurrence index --
        // as many as the most frequent label has rows in the group -- were measured: no faster.)
        auto clash_mask = [&](int lab) -> unsigned long long {
            if (RPI == 1) return 0ull;
            const int base = lane / RPI * RPI, pos = lane - base;
            bool cl = false;
#pragma unroll
            for (int d = 1; d < RPI; d++) {
                const int p = pos + d < RPI ? pos + d : pos + d - RPI;
                cl = cl | (__shfl(lab, base + p) == lab);  // every lane takes part in every exchange
            }
            return __ballot(cl && lab != k && lane < RL);
        };
        bits_t val[U];
        int lv_cur[NL], lv_nxt[NL], lv_far[NL];   // labels two tiles ahead, requested before the tile's values
        load_labels(0, lv_cur);
        load_labels(TR, lv_nxt);
        {
            unsigned off = 0;
#pragma unroll
            for (int g = 0; g < U; g++, off += gstep) val[g] = load_val(g * RPI, off);
        }
        // byte address of this lane's pair in the table row of a label: tbl + (label * c + 2 pr) * 8
        char *const lane_word = reinterpret_cast<char *>(tbl) + (active ? 2 * pr : (k + 1) * c + 2 * lane) * 8;
        unsigned tile_off = TR * gstep / RPI;
        for (int rel0 = 0; rel0 < lim; rel0 += TR, tile_off += TR * gstep / RPI) {
            load_labels(rel0 + 2 * TR, lv_far);
            unsigned long long cm[NL];
            int row_bytes[NL];
#pragma unroll
            for (int i = 0; i < NL; i++) {
                if (lv_cur[i] < k) atomicAdd(&cnt[lv_cur[i]], 1u);
                cm[i] = clash_mask(lv_cur[i]);
                row_bytes[i] = (int)__umul24(lv_cur[i], c * 8);
            }
            int word[U];
#pragma unroll
            for (int g = 0; g < U; g++) {
                const int r = g * RPI;
                const int rb8 = __shfl(row_bytes[r / RL], r % RL + slot);
                word[g] = active ? rb8 : 0;
            }
            unsigned off = tile_off;
#pragma unroll
            for (int g = 0; g < U; g++) {
                d2 *const wp = reinterpret_cast<d2 *>(lane_word + word[g]);
                const d2 v = PairBits<T>::widen(val[g]);
                val[g] = load_val(rel0 + TR + g * RPI, off);  // this register's load for the next tile
                off += gstep;
                if (!((cm[g * RPI / RL] >> (g * RPI % RL)) & ((1ull << RPI) - 1))) {
                    *wp = *wp + v;
                } else {
                    // one row at a time.  The fences keep the RPI predicated updates apart: to the compiler they
                    // are mutually exclusive branches of one thread, which it may fold into a single update
#pragma unroll
                    for (int s = 0; s < RPI; s++) {
                        if (slot == s) *wp = *wp + v;
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

template <typename T, int RPI, bool COUNT_F64>
bool launch_pairs(const T *x, int64_t n, int c, int64_t ldx, const int32_t *labels, int k, double *sums, void *counts,
                  hipStream_t st, int nwv, int blocks_per_cu)
{
    const size_t tbytes = ((size_t)(k + 1) * c + 128) * 8;
    const size_t lds = tbytes * nwv + (size_t)k * 4;
    if (lds > 159 * 1024) return false;
    constexpr int TR = RPI * (sizeof(T) == 8 ? 16 : 32);
    const int64_t max_waves = (int64_t)device_cu_count() * blocks_per_cu * nwv;
    int64_t rows_per_wave = (n + max_waves - 1) / max_waves;
    if (rows_per_wave < 4 * TR) rows_per_wave = 4 * TR;
    rows_per_wave = (rows_per_wave + TR - 1) / TR * TR;
    if ((rows_per_wave + 1024) * ldx * (int64_t)sizeof(T) >= (1ll << 31)) return false;   // 32-bit buffer offsets
    const int64_t waves = (n + rows_per_wave - 1) / rows_per_wave;
    const int64_t grid = (waves + nwv - 1) / nwv;
    auto kern = cluster_sums_pairs_kernel<T, RPI, COUNT_F64>;
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess)
        return false;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * nwv), lds, st, x, n, c, ldx, labels, k, sums,
                       reinterpret_cast<unsigned long long *>(counts), rows_per_wave);
    return true;
}

template <typename T, int RPI>
bool launch_pairs_counts(const T *x, int64_t n, int c, int64_t ldx, const int32_t *labels, int k, double *sums,
                         void *counts, bool counts_f64, hipStream_t st, int nwv, int blocks_per_cu)
{
    return counts_f64 ? launch_pairs<T, RPI, true>(x, n, c, ldx, labels, k, sums, counts, st, nwv, blocks_per_cu)
                      : launch_pairs<T, RPI, false>(x, n, c, ldx, labels, k, sums, counts, st, nwv, blocks_per_cu);
}

}  // namespace

template <typename T>
bool launch_sums_pairs(const T *x, int64_t n, int c, int64_t ldx, const int32_t *labels, int k, double *sums,
                       void *counts, bool counts_f64, hipStream_t st, int nwv, int blocks_per_cu)
{
    if (c % 2 || ldx % 2 || c < 14 || c > 64 || reinterpret_cast<uintptr_t>(x) % (2 * sizeof(T))) return false;
    const int rpi = 64 / (c / 2);
#define PXSOM_PAIRS(R) return launch_pairs_counts<T, R>(x, n, c, ldx, labels, k, sums, counts, counts_f64, st, nwv, blocks_per_cu)
    if (rpi >= 8) PXSOM_PAIRS(8);
    if (rpi >= 5) PXSOM_PAIRS(5);
    if (rpi == 4) PXSOM_PAIRS(4);
    if (rpi == 3) PXSOM_PAIRS(3);
    PXSOM_PAIRS(2);
#undef PXSOM_PAIRS
}

template bool launch_sums_pairs<float>(const float *, int64_t, int, int64_t, const int32_t *, int, double *, void *, bool,
                                       hipStream_t, int, int);
template bool launch_sums_pairs<_Float16>(const _Float16 *, int64_t, int, int64_t, const int32_t *, int, double *, void *,
                                          bool, hipStream_t, int, int);
template bool launch_sums_pairs<double>(const double *, int64_t, int, int64_t, const int32_t *, int, double *, void *, bool,
                                        hipStream_t, int, int);

}  // namespace pxsom
