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

#include "pxsom_common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kHdrBytes = 256;
constexpr int kTilesPerIter = 4;  // 4 tiles x 16 pixels = one 64-row group per wave iteration
constexpr float kNegBig = -3.0e38f;

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
};

struct Layout {
    int nb, nch, cpl, nsteps, idx_bits;
    size_t off_wfrag, off_bias, off_list, total;
};

inline Layout make_layout(int64_t n, int c, int k)
{
    Layout L;
    L.nb = (k + 15) / 16;
    L.nch = (c + 31) / 32;
    int per_chunk = (c + L.nch - 1) / L.nch;          // channels per chunk
    int cpl = (per_chunk + 3) / 4;                     // per lane (4 lane groups)
    cpl = (cpl + 1) & ~1;                              // even, so float2/double2 loads stay aligned
    if (cpl > 8) cpl = 8;
    L.cpl = cpl;
    L.nsteps = 2 * L.nch;  // stored fragments per node block: {Wh, Wl} per chunk
    L.idx_bits = L.nb <= 16 ? 6 : 10;
    L.off_wfrag = kHdrBytes;
    L.off_bias = L.off_wfrag + (size_t)L.nb * L.nsteps * 64 * sizeof(half8);
    L.off_list = pxsom::align_up(L.off_bias + (size_t)L.nb * 64 * sizeof(f32x4), 256);
    L.total = L.off_list + (size_t)(n > 0 ? n : 1) * sizeof(unsigned);
    return L;
}

// ------------------------------------------------------------------------------------------------
// 1. prep: one workgroup of 256 threads.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bmu_prep_kernel(const double *__restrict__ w, int k, int c,
                                                       AssignHdr *hdr, half8 *wfrag, f32x4 *bias,
                                                       int nb, int nch, int cpl, int idx_bits)
{
    __shared__ double s_norm2[PXSOM_MAX_NODES];
    __shared__ double s_red[256];
    __shared__ int s_bad;
    const int tid = threadIdx.x;
    if (tid == 0) s_bad = 0;
    __syncthreads();

    // per-node squared norm (binary64) and global max |w|
    double mymax = 0.0;
    for (int node = tid; node < k; node += 256) {
        double s = 0.0;
        for (int j = 0; j < c; j++) {
            double v = w[(size_t)node * c + j];
            if (!(fabs(v) <= DBL_MAX)) s_bad = 1;  // NaN / Inf in the codebook
            s += v * v;
            mymax = fmax(mymax, fabs(v));
        }
        s_norm2[node] = s;
    }
    s_red[tid] = mymax;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) s_red[tid] = fmax(s_red[tid], s_red[tid + off]);
        __syncthreads();
    }
    const double maxabs = s_red[0];
    // scale = 2^e with maxabs*scale in [128, 256): fp16 keeps 11 significant bits there and the
    // low halves of the split stay normal down to |x| ~ 1e-4 * maxabs.
    int e = 0;
    if (maxabs > 0.0 && maxabs <= DBL_MAX) {
        int ex;
        frexp(maxabs, &ex);  // maxabs = m * 2^ex, m in [0.5, 1)
        e = 8 - ex;
        if (e > 100) e = 100;
        if (e < -100) e = -100;
    }
    const double scale = ldexp(1.0, e);

    double wn2max = 0.0;
    for (int node = tid; node < k; node += 256) wn2max = fmax(wn2max, s_norm2[node]);
    __syncthreads();
    s_red[tid] = wn2max;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) s_red[tid] = fmax(s_red[tid], s_red[tid + off]);
        __syncthreads();
    }
    if (tid == 0) {
        const bool bad = s_bad != 0 || !(s_red[0] * scale * scale <= 1.0e30);
        hdr->amb_count = 0;
        hdr->scale = (float)scale;
        // rounded up by a hair; an infinite wn_max makes every row take the exact path
        hdr->wn_max = bad ? INFINITY : (float)(sqrt(s_red[0]) * scale * (1.0 + 1e-6));
        // coefficient of the rigorous |filter - exact| bound, see DESIGN.md "K7 error bound":
        //   index packing 2^-(23-idx_bits), fp32 accumulation (3C+2)*2^-24, split residual 2^-19,
        //   f64->f32 input rounding 2^-23;  tol = 2 * 1.25 * E
        const double coef = ldexp(1.0, -(23 - idx_bits)) + (3.0 * c + 2.0) * ldexp(1.0, -24) +
                            ldexp(1.0, -19) + ldexp(1.0, -23);
        hdr->tol_rel = (float)(2.5 * coef);
        hdr->tol_abs = (float)(2.5 * ldexp(1.0, -24) * sqrt((double)c));  // fp16 subnormal floor
        hdr->x_limit = 60000.0f;
        hdr->nb = nb;
        hdr->nch = nch;
        hdr->cpl = cpl;
        hdr->idx_bits = idx_bits;
    }

    // A-fragments: wfrag[(b*nsteps + s)*64 + lane], lane = (q<<4 | m): node 16b+m,
    // slot i of lane group q in chunk h <-> channel h*4*cpl + q*cpl + i (i < cpl)
    const int nsteps = 2 * nch;
    for (int f = tid; f < nb * nsteps * 64; f += 256) {
        const int lane = f & 63, s = (f >> 6) % nsteps, b = (f >> 6) / nsteps;
        const int m = lane & 15, q = lane >> 4, h = s / 2, t = s % 2;
        const int node = b * 16 + m;
        half8 frag;
        for (int i = 0; i < 8; i++) {
            const int ch = h * 4 * cpl + q * cpl + i;
            float W = 0.f;
            if (i < cpl && ch < c && node < k) W = (float)(w[(size_t)node * c + ch] * scale);
            const _Float16 hi = (_Float16)W;
            const _Float16 lo = (_Float16)(W - (float)hi);
            frag[i] = (t == 1) ? lo : hi;
        }
        wfrag[f] = frag;
    }
    // bias[b*64 + lane][r] for accumulator row (lane>>4)*4 + r <-> node 16b + 4q + r
    for (int f = tid; f < nb * 64; f += 256) {
        const int lane = f & 63, b = f >> 6, q = lane >> 4;
        f32x4 bv;
        for (int r = 0; r < 4; r++) {
            const int node = b * 16 + q * 4 + r;
            bv[r] = node < k ? (float)(-0.5 * s_norm2[node] * scale * scale) : kNegBig;
        }
        bias[f] = bv;
    }
}

// ------------------------------------------------------------------------------------------------
// 2. filter
// ------------------------------------------------------------------------------------------------
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

__device__ __forceinline__ float pack_idx(float v, unsigned idx, unsigned mask)
{
    return __uint_as_float((__float_as_uint(v) & ~mask) | idx);
}

// NB_T > 0: codebook fragments + bias live in registers (NCH_T*2*NB_T*4 + NB_T*4 VGPRs);
// NB_T == 0: fragments are streamed from the workspace (L1/L2 resident), any K.
// CPL_T > 0: compile-time channels-per-lane; 0: runtime.
// PREFETCH: the next 64-row group's loads are issued before the current group's MFMA work.
template <typename T, int NCH_T, int CPL_T, int NB_T, bool VEC2, bool PREFETCH>
__global__ __launch_bounds__(256, 2) void bmu_filter_kernel(
    const T *__restrict__ x, int64_t n, int c, int64_t ldx, const half8 *__restrict__ wfrag,
    const f32x4 *__restrict__ bias, AssignHdr *hdr, unsigned *__restrict__ amb_list,
    int32_t *__restrict__ labels)
{
    constexpr int NCH = NCH_T;
    constexpr int NFR = 2 * NCH;  // stored fragments per node block
    constexpr int CPLMAX = CPL_T > 0 ? CPL_T : 8;
    const int cpl = CPL_T > 0 ? CPL_T : hdr->cpl;
    const int nb = NB_T > 0 ? NB_T : hdr->nb;
    const unsigned idx_mask = NB_T > 0 ? 63u : ((1u << hdr->idx_bits) - 1u);
    const float scale = hdr->scale, wn_max = hdr->wn_max, tol_rel = hdr->tol_rel,
                tol_abs = hdr->tol_abs, x_limit = hdr->x_limit;

    const int lane = threadIdx.x & 63;
    const int pix = lane & 15, q = lane >> 4;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
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

    // dst[h][i]: channel h*4*cpl + q*cpl + i of row g*64 + t*16 + pix
    auto load_tile = [&](int64_t g, int t, T(&dst)[NCH][CPLMAX]) {
        const int64_t row = g * 64 + t * 16 + pix;
        const bool rok = row < n;
        const T *rp = x + row * ldx;
#pragma unroll
        for (int h = 0; h < NCH; h++) {
            const int ch0 = h * 4 * cpl + q * cpl;
            if constexpr (VEC2) {
#pragma unroll
                for (int i = 0; i < CPLMAX; i += 2) {
                    typename Pair<T>::type v;
                    v.x = (T)0;
                    v.y = (T)0;
                    if (rok && i < cpl && ch0 + i < c)
                        v = *reinterpret_cast<const typename Pair<T>::type *>(rp + ch0 + i);
                    dst[h][i] = v.x;
                    dst[h][i + 1] = v.y;
                }
            } else {
#pragma unroll
                for (int i = 0; i < CPLMAX; i++) {
                    T v = (T)0;
                    if (rok && i < cpl && ch0 + i < c) v = rp[ch0 + i];
                    dst[h][i] = v;
                }
            }
        }
    };

    T raw[PREFETCH ? kTilesPerIter : 1][NCH][CPLMAX];

    int64_t g = wave;
    if constexpr (PREFETCH) {
        if (g < ngroups) {
#pragma unroll
            for (int t = 0; t < kTilesPerIter; t++) load_tile(g, t, raw[t]);
        }
    }
    for (; g < ngroups; g += nwaves) {
        half8 bh[PREFETCH ? kTilesPerIter : 1][NCH], bl[PREFETCH ? kTilesPerIter : 1][NCH];
        float ss[PREFETCH ? kTilesPerIter : 1];
        auto convert = [&](int slot) {
            float acc2 = 0.f;
#pragma unroll
            for (int h = 0; h < NCH; h++) {
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    float xf = 0.f;
                    if (i < CPLMAX) xf = (float)raw[slot][h][i < CPLMAX ? i : 0] * scale;
                    acc2 = fmaf(xf, xf, acc2);
                    const _Float16 hi = (_Float16)xf;
                    bh[slot][h][i] = hi;
                    bl[slot][h][i] = (_Float16)(xf - (float)hi);
                }
            }
            ss[slot] = acc2;
        };
        if constexpr (PREFETCH) {
            // convert the current group's rows to fp16 hi/lo B-fragments, then prefetch the next
#pragma unroll
            for (int t = 0; t < kTilesPerIter; t++) convert(t);
            const int64_t gnext = g + nwaves;
            if (gnext < ngroups) {
#pragma unroll
                for (int t = 0; t < kTilesPerIter; t++) load_tile(gnext, t, raw[t]);
            }
        }

        int my_label = 0;
        bool my_amb = false;
#pragma unroll
        for (int t = 0; t < kTilesPerIter; t++) {
            const int slot = PREFETCH ? t : 0;
            if constexpr (!PREFETCH) {
                load_tile(g, t, raw[0]);
                convert(0);
            }
            float m1 = kNegBig, m2 = kNegBig;
            auto consume = [&](const f32x4 &acc, int b) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float v = pack_idx(acc[r], (unsigned)(b * 4 + r), idx_mask);
                    m2 = __builtin_amdgcn_fmed3f(m1, m2, v);
                    m1 = fmaxf(m1, v);
                }
            };
            // per chunk: Wh*Xh + Wh*Xl + Wl*Xh
            if constexpr (NB_T > 0) {
#pragma unroll
                for (int b = 0; b < NB_T; b++) {
                    f32x4 acc = breg[b];
#pragma unroll
                    for (int h = 0; h < NCH; h++) {
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[b][2 * h], bh[slot][h], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[b][2 * h], bl[slot][h], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[b][2 * h + 1], bh[slot][h], acc, 0, 0, 0);
                    }
                    consume(acc, b);
                }
            } else {
                for (int b = 0; b < nb; b++) {
                    f32x4 acc = bias[b * 64 + lane];
#pragma unroll
                    for (int h = 0; h < NCH; h++) {
                        const half8 wh = wfrag[(b * NFR + 2 * h) * 64 + lane];
                        const half8 wl = wfrag[(b * NFR + 2 * h + 1) * 64 + lane];
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, bh[slot][h], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, bl[slot][h], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, bh[slot][h], acc, 0, 0, 0);
                    }
                    consume(acc, b);
                }
            }
            // packed register index -> node index 16b + 4q + r
            const unsigned idx = __float_as_uint(m1) & idx_mask;
            int node = (int)(((idx >> 2) << 4) | ((unsigned)q << 2) | (idx & 3u));
            float s2 = ss[slot];
            // merge the 4 lane groups that share this pixel (lanes pix, pix+16, pix+32, pix+48)
#pragma unroll
            for (int off = 16; off <= 32; off <<= 1) {
                const float o1 = __shfl_xor(m1, off);
                const float o2 = __shfl_xor(m2, off);
                const int on = __shfl_xor(node, off);
                s2 += __shfl_xor(s2, off);
                const bool take = (o1 > m1) || (o1 == m1 && on < node);
                m2 = fmaxf(fminf(m1, o1), fmaxf(m2, o2));
                m1 = take ? o1 : m1;
                node = take ? on : node;
            }
            const float xn = sqrtf(s2);
            const float tol = tol_rel * (xn * wn_max + 0.5f * wn_max * wn_max) + tol_abs * (xn + wn_max);
            // NaN-safe: anything not provably unique goes to the exact path
            const bool amb = !((m1 - m2) > tol) || !(xn < x_limit);
            if (q == t) {
                my_label = node + 1;
                my_amb = amb;
            }
        }
        // lane (q, pix) now owns row g*64 + q*16 + pix == g*64 + lane
        const int64_t row = g * 64 + lane;
        const bool valid = row < n;
        if (valid) labels[row] = my_label;
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

// ------------------------------------------------------------------------------------------------
// 3. exact path: one wave per listed row; lane <-> nodes lane, lane+64, ...
//    binary64, no contraction, j ascending, sqrt, first strict minimum (FlowSOM C_mapDataToCodes).
// ------------------------------------------------------------------------------------------------
#pragma clang fp contract(off)
template <typename T>
__global__ __launch_bounds__(256) void bmu_exact_kernel(const T *__restrict__ x, int c, int64_t ldx,
                                                        const double *__restrict__ w, int k,
                                                        const AssignHdr *hdr,
                                                        const unsigned *__restrict__ amb_list,
                                                        int32_t *__restrict__ labels, int use_lds)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double *wt = reinterpret_cast<double *>(smem_raw);  // [c][k] transposed codebook
    const unsigned count = hdr->amb_count;
    if (blockIdx.x * 4u >= count) return;  // uniform per workgroup: nothing listed for it
    if (use_lds) {
        for (int e = threadIdx.x; e < k * c; e += 256) {
            const int node = e / c, j = e - node * c;
            wt[(size_t)j * k + node] = w[e];
        }
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    const unsigned wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    for (unsigned e = wave; e < count; e += nwaves) {
        const int64_t row = amb_list[e];
        const T *rp = x + row * ldx;
        double best = DBL_MAX;
        int bestk = 0x7fffffff;
        for (int node = lane; node < k; node += 64) {
            double xdist = 0.0;
            for (int j = 0; j < c; j++) {
                const double wv = use_lds ? wt[(size_t)j * k + node] : w[(size_t)node * c + j];
                const double tmp = (double)rp[j] - wv;
                xdist += tmp * tmp;
            }
            const double d = sqrt(xdist);
            if (d < best) {
                best = d;
                bestk = node;
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double od = __shfl_xor(best, off);
            const int ok = __shfl_xor(bestk, off);
            if (od < best || (od == best && ok < bestk)) {
                best = od;
                bestk = ok;
            }
        }
        if (lane == 0) labels[row] = bestk == 0x7fffffff ? 0 : bestk + 1;
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

template <typename T, int NCH, int CPL, int NB, bool VEC2>
void launch_filter(const T *x, int64_t n, int c, int64_t ldx, char *ws, const Layout &L, int32_t *labels,
                   hipStream_t st, int grid)
{
    hipLaunchKernelGGL((bmu_filter_kernel<T, NCH, CPL, NB, VEC2, (NCH == 1)>), dim3(grid), dim3(256), 0, st, x, n, c,
                       ldx, reinterpret_cast<const half8 *>(ws + L.off_wfrag),
                       reinterpret_cast<const f32x4 *>(ws + L.off_bias), reinterpret_cast<AssignHdr *>(ws),
                       reinterpret_cast<unsigned *>(ws + L.off_list), labels);
}

template <typename T>
int assign_typed(const T *x, int64_t n, int c, int64_t ldx, const double *w, int k, int32_t *labels,
                 double *dist, char *ws, const Layout &L, hipStream_t st)
{
    hipLaunchKernelGGL(bmu_prep_kernel, dim3(1), dim3(256), 0, st, w, k, c, reinterpret_cast<AssignHdr *>(ws),
                       reinterpret_cast<half8 *>(ws + L.off_wfrag), reinterpret_cast<f32x4 *>(ws + L.off_bias),
                       L.nb, L.nch, L.cpl, L.idx_bits);
    PXSOM_LAUNCH_CHECK("bmu_prep_kernel");

    const int cus = pxsom::device_cu_count();
    const int64_t ngroups = (n + 63) / 64;
    int grid = (int)std::min<int64_t>((ngroups + 3) / 4, (int64_t)cus * 2);
    if (grid < 1) grid = 1;
    // pair loads need 2-element alignment of every row start and of the base pointer
    const bool vec2 = (c % 2 == 0) && (ldx % 2 == 0) &&
                      (reinterpret_cast<uintptr_t>(x) % (2 * sizeof(T)) == 0);
    // headline shape (BASELINE.json configs 2/3: C=22, K=100): register-resident codebook
    if (vec2 && L.nch == 1 && L.cpl == 6 && L.nb == 7)
        launch_filter<T, 1, 6, 7, true>(x, n, c, ldx, ws, L, labels, st, grid);
    else if (vec2 && L.nch == 1 && L.cpl == 2 && L.nb == 7)  // config 1 (C=8, K=100)
        launch_filter<T, 1, 2, 7, true>(x, n, c, ldx, ws, L, labels, st, grid);
    else if (L.nch == 1)
        vec2 ? launch_filter<T, 1, 0, 0, true>(x, n, c, ldx, ws, L, labels, st, grid)
             : launch_filter<T, 1, 0, 0, false>(x, n, c, ldx, ws, L, labels, st, grid);
    else if (L.nch == 2)
        vec2 ? launch_filter<T, 2, 0, 0, true>(x, n, c, ldx, ws, L, labels, st, grid)
             : launch_filter<T, 2, 0, 0, false>(x, n, c, ldx, ws, L, labels, st, grid);
    else if (L.nch == 3)
        vec2 ? launch_filter<T, 3, 0, 0, true>(x, n, c, ldx, ws, L, labels, st, grid)
             : launch_filter<T, 3, 0, 0, false>(x, n, c, ldx, ws, L, labels, st, grid);
    else
        vec2 ? launch_filter<T, 4, 0, 0, true>(x, n, c, ldx, ws, L, labels, st, grid)
             : launch_filter<T, 4, 0, 0, false>(x, n, c, ldx, ws, L, labels, st, grid);
    PXSOM_LAUNCH_CHECK("bmu_filter_kernel");

    const size_t wt_bytes = (size_t)k * c * sizeof(double);
    const int use_lds = wt_bytes <= 64 * 1024;
    int egrid = (int)std::min<int64_t>((n + 3) / 4, (int64_t)cus * 4);
    if (egrid < 1) egrid = 1;
    hipLaunchKernelGGL(bmu_exact_kernel<T>, dim3(egrid), dim3(256), use_lds ? wt_bytes : 0, st, x, c, ldx, w,
                       k, reinterpret_cast<const AssignHdr *>(ws),
                       reinterpret_cast<const unsigned *>(ws + L.off_list), labels, use_lds);
    PXSOM_LAUNCH_CHECK("bmu_exact_kernel");

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
    if (n < 0 || n > 0x7fffffffLL)
        return pxsom::fail(PXSOM_ERR_INVALID_ARG, "pxsom_assign: n=%lld outside [0, 2^31)", (long long)n);
    if (c < 1 || c > PXSOM_MAX_CHANNELS)
        return pxsom::fail(PXSOM_ERR_UNSUPPORTED, "pxsom_assign: c=%d outside [1, %d]", c, PXSOM_MAX_CHANNELS);
    if (k < 1 || k > PXSOM_MAX_NODES)
        return pxsom::fail(PXSOM_ERR_UNSUPPORTED, "pxsom_assign: k=%d outside [1, %d]", k, PXSOM_MAX_NODES);
    if (ldx < c) return pxsom::fail(PXSOM_ERR_INVALID_ARG, "pxsom_assign: ldx=%lld < c=%d", (long long)ldx, c);
    if (dtype != PXSOM_F32 && dtype != PXSOM_F64)
        return pxsom::fail(PXSOM_ERR_UNSUPPORTED, "pxsom_assign: dtype %d", dtype);
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
    if (dtype == PXSOM_F32)
        return assign_typed<float>(reinterpret_cast<const float *>(x_dev), n, c, ldx, w_dev, k, labels_dev,
                                   dist_dev, ws, L, st);
    return assign_typed<double>(reinterpret_cast<const double *>(x_dev), n, c, ldx, w_dev, k, labels_dev,
                                dist_dev, ws, L, st);
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
