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

#include "pxsom_assign.h"

using namespace pxsom_bmu;

namespace {

// ------------------------------------------------------------------------------------------------
// 1. prep: one workgroup of 256 threads.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bmu_prep_kernel(const double *__restrict__ w, int k, int c,
                                                       AssignHdr *hdr, half8 *wfrag, f32x4 *bias,
                                                       int nb, int nch, int cpl, int idx_bits,
                                                       int node_bits, int stage)
{
    __shared__ double s_norm2[PXSOM_MAX_NODES];
    __shared__ double s_red[256];
    __shared__ int s_bad;
    extern __shared__ __attribute__((aligned(16))) char prep_smem[];
    const int tid = threadIdx.x;
    if (tid == 0) s_bad = 0;
    // small codebooks are staged in LDS with one coalesced sweep; every later read is an LDS read
    const double *wl = w;
    if (stage) {
        double *sw = reinterpret_cast<double *>(prep_smem);
        for (int e = tid; e < k * c; e += 256) sw[e] = w[e];
        wl = sw;
    }
    __syncthreads();

    // per-node squared norm (binary64) and global max |w|
    double mymax = 0.0;
    for (int node = tid; node < k; node += 256) {
        double s = 0.0;
        for (int j = 0; j < c; j++) {
            double v = wl[(size_t)node * c + j];
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
        hdr->wn_max = bad ? 0.f : (float)(sqrt(s_red[0]) * scale * (1.0 + 1e-6));
        hdr->force_exact = bad ? 1 : 0;  // NaN/Inf/huge codebook: every row takes the exact path
        // coefficient of the rigorous |filter - exact| bound, see DESIGN.md "K7 error bound":
        //   index packing 2^-(23-idx_bits) (idx_bits low mantissa bits replaced), fp32 accumulation
        //   (3C+2)*2^-24, split residual 2^-19,
        //   f64->f32 input rounding 2^-23;  tol = 2 * 1.25 * E
        const double coef = ldexp(1.0, -(23 - idx_bits)) +
                            (3.0 * c + 2.0) * ldexp(1.0, -24) + ldexp(1.0, -19) + ldexp(1.0, -23);
        hdr->tol_rel = (float)(2.5 * coef);
        hdr->tol_abs = (float)(2.5 * ldexp(1.0, -24) * sqrt((double)c));  // fp16 subnormal floor
        hdr->x_limit = 60000.0f;
        hdr->nb = nb;
        hdr->nch = nch;
        hdr->cpl = cpl;
        hdr->idx_bits = idx_bits;
        hdr->node_bits = node_bits;
    }

    // A-fragments: wfrag[(b*nsteps + s)*64 + lane], lane = (q<<4 | m): node 16b+m,
    // slot i of lane group q in chunk h <-> channel h*4*cpl + q*cpl + i (i < cpl)
    const int nsteps = 2 * nch;
    for (int f = tid; f < nb * nsteps * 64; f += 256) {
        const int lane = f & 63, s = (f >> 6) % nsteps, b = (f >> 6) / nsteps;
        const int m = lane & 15, q = lane >> 4, h = s / 2, t = s % 2;
        const int node = node_of_row(b, m, nb);
        half8 frag;
        for (int i = 0; i < 8; i++) {
            const int ch = h * 4 * cpl + q * cpl + i;
            float W = 0.f;
            if (i < cpl && ch < c && node < k) W = (float)(wl[(size_t)node * c + ch] * scale);
            const _Float16 hi = (_Float16)W;
            const _Float16 lo = (_Float16)(W - (float)hi);
            frag[i] = (t == 1) ? lo : hi;
        }
        wfrag[f] = frag;
    }
    // Exact duplicates of an EARLIER node can never be the answer (their distance is identical and the
    // reference keeps the first minimum), so they are masked out of the filter.  This matters in batch
    // training: while the neighbourhood radius still spans the grid, all central nodes receive the same
    // update and are bit-identical, which would otherwise send every row they win to the exact path.
    // (s_dup reuses s_red's storage class: one flag per node.)
    __shared__ unsigned char s_dup[PXSOM_MAX_NODES];
    for (int node = tid; node < k; node += 256) s_dup[node] = 0;
    __syncthreads();
    // hash table keyed by the norm's bit pattern (equal rows have equal norms): slot <- smallest node
    // index hashing there; a node is a duplicate iff an earlier node with identical channels exists.
    {
        __shared__ int s_tab[1024];
        for (int i = tid; i < 1024; i += 256) s_tab[i] = 0x7fffffff;
        __syncthreads();
        auto slot_of = [&](int node) {
            const unsigned long long b = (unsigned long long)__double_as_longlong(s_norm2[node]);
            return (int)((b ^ (b >> 17) ^ (b >> 41)) & 1023ull);
        };
        for (int node = tid; node < k; node += 256) atomicMin(&s_tab[slot_of(node)], node);
        __syncthreads();
        for (int node = tid; node < k; node += 256) {
            const int first = s_tab[slot_of(node)];
            if (first >= node) continue;
            auto same_as = [&](int prev) {
                if (s_norm2[prev] != s_norm2[node]) return false;
                for (int j = 0; j < c; j++)
                    if (wl[(size_t)prev * c + j] != wl[(size_t)node * c + j]) return false;
                return true;
            };
            bool dup = same_as(first);
            if (!dup) {  // slot shared with a different earlier node: scan the norms (no early exit, so
                         // the LDS reads pipeline), full comparison only on a norm match
                const double n2 = s_norm2[node];
                int hit = -1;
#pragma unroll 8
                for (int prev = 0; prev < node; prev++)
                    if (s_norm2[prev] == n2 && prev != first && hit < 0) hit = prev;
                if (hit >= 0) {
                    for (int prev = hit; prev < node && !dup; prev++) dup = same_as(prev);
                }
            }
            if (dup) s_dup[node] = 1;
        }
    }
    __syncthreads();
    // bias[b*64 + lane][r] for accumulator row (lane>>4)*4 + r <-> node_of_row(b, 4q + r)
    for (int f = tid; f < nb * 64; f += 256) {
        const int lane = f & 63, b = f >> 6, q = lane >> 4;
        f32x4 bv;
        for (int r = 0; r < 4; r++) {
            const int node = node_of_row(b, q * 4 + r, nb);
            bv[r] = (node < k && !s_dup[node]) ? (float)(-0.5 * s_norm2[node] * scale * scale) : kNegBig;
        }
        bias[f] = bv;
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
    // RB listed rows per wave iteration: the codebook element read from LDS is shared by the RB rows
    // and their independent binary64 chains hide each other's latency.  Each row is fetched once
    // (lane j holds channels j and j+64) and broadcast with v_readlane (j is wave-uniform).
    constexpr int RB = 4;
    for (unsigned e0 = wave * RB; e0 < count; e0 += nwaves * RB) {
        int64_t rows[RB];
        unsigned x_lo[RB][2], x_hi[RB][2];
#pragma unroll
        for (int u = 0; u < RB; u++) {
            const unsigned e = e0 + u < count ? e0 + u : count - 1;  // surplus slots redo the last row
            rows[u] = amb_list[e];
            const T *rp = x + rows[u] * ldx;
            const double xa = lane < c ? (double)rp[lane] : 0.0;
            const double xb = lane + 64 < c ? (double)rp[lane + 64] : 0.0;
            x_lo[u][0] = (unsigned)__double_as_longlong(xa);
            x_hi[u][0] = (unsigned)(__double_as_longlong(xa) >> 32);
            x_lo[u][1] = (unsigned)__double_as_longlong(xb);
            x_hi[u][1] = (unsigned)(__double_as_longlong(xb) >> 32);
        }
        double best[RB];
        int bestk[RB];
#pragma unroll
        for (int u = 0; u < RB; u++) {
            best[u] = DBL_MAX;
            bestk[u] = 0x7fffffff;
        }
        for (int base = 0; base < k; base += 128) {
            const int n0 = base + lane, n1 = base + 64 + lane;
            const int c0 = n0 < k ? n0 : k - 1, c1 = n1 < k ? n1 : k - 1;
            double d0[RB], d1[RB];
#pragma unroll
            for (int u = 0; u < RB; u++) d0[u] = d1[u] = 0.0;
            for (int j = 0; j < c; j++) {
                const double w0 = use_lds ? wt[(size_t)j * k + c0] : w[(size_t)c0 * c + j];
                const double w1 = use_lds ? wt[(size_t)j * k + c1] : w[(size_t)c1 * c + j];
                const int h = j >> 6, jj = j & 63;
#pragma unroll
                for (int u = 0; u < RB; u++) {
                    const unsigned lo = __builtin_amdgcn_readlane(h ? x_lo[u][1] : x_lo[u][0], jj);
                    const unsigned hi = __builtin_amdgcn_readlane(h ? x_hi[u][1] : x_hi[u][0], jj);
                    const double xj = __longlong_as_double(((long long)hi << 32) | lo);
                    const double t0 = xj - w0, t1 = xj - w1;
                    d0[u] += t0 * t0;
                    d1[u] += t1 * t1;
                }
            }
#pragma unroll
            for (int u = 0; u < RB; u++) {
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
        for (int u = 0; u < RB; u++) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const double od = __shfl_xor(best[u], off);
                const int ok = __shfl_xor(bestk[u], off);
                if (od < best[u] || (od == best[u] && ok < bestk[u])) {
                    best[u] = od;
                    bestk[u] = ok;
                }
            }
            if (lane == 0) labels[rows[u]] = bestk[u] == 0x7fffffff ? 0 : bestk[u] + 1;
        }
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

template <typename T>
int assign_typed(const T *x, int64_t n, int c, int64_t ldx, const double *w, int k, int32_t *labels,
                 double *dist, char *ws, const Layout &L, hipStream_t st)
{
    const size_t stage_bytes = (size_t)k * c * sizeof(double);
    const int stage = stage_bytes <= 40 * 1024;
    hipLaunchKernelGGL(bmu_prep_kernel, dim3(1), dim3(256), stage ? stage_bytes : 0, st, w, k, c,
                       reinterpret_cast<AssignHdr *>(ws),
                       reinterpret_cast<half8 *>(ws + L.off_wfrag), reinterpret_cast<f32x4 *>(ws + L.off_bias),
                       L.nb, L.nch, L.cpl, L.idx_bits, L.node_bits, stage);
    PXSOM_LAUNCH_CHECK("bmu_prep_kernel");

    const int cus = pxsom::device_cu_count();
    pxsom::Prof *prof = pxsom::current_prof();
    pxsom::prof_mark(prof, st, true, n);
    launch_filter_any<T>(x, n, c, ldx, ws, L, labels, st);
    pxsom::prof_mark(prof, st, false, n);
    PXSOM_LAUNCH_CHECK("bmu_filter_kernel");

    const size_t wt_bytes = (size_t)k * c * sizeof(double);
    const int use_lds = wt_bytes <= 64 * 1024;
    // listed rows are a small fraction of n; the kernel grid-strides over the list anyway
    int egrid = (int)std::min<int64_t>((n + 255) / 256, (int64_t)cus * 4);
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
