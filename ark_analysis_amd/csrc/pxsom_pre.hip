// pxsom_pre.hip -- pixel-matrix pre-processing on gfx950 (create_fov_pixel_data and the 99.9 % values).
//
//   pxsom_gaussian_blur_hwc        scipy.ndimage.gaussian_filter(plane, sigma) per channel
//                                  (/root/reference/src/ark/phenotyping/pixie_preprocessing.py:47-49)
//   pxsom_rowsum_filter_normalize  row-sum threshold + non-zero filter + row normalisation + compaction
//                                  (pixie_preprocessing.py:67-75, pixel_cluster_utils.py:126-130)
//   pxsom_normalize_columns        x[:, j] / norm[j]  (cluster_helpers.py:242-246)
//   pxsom_quantile_nonzero         type-7 quantile of the non-zero values of each column
//                                  (pixie_preprocessing.py:406-408, cluster_helpers.py:366,
//                                   pixel_cluster_utils.py:47-51)
// All binary64 with one rounding per operation in the reference's order (fp contraction off), so the
// results equal the oracle's, which is pinned to scipy / pandas / numpy (tests/test_oracle_golden.py).
// These kernels are HBM-streaming: every element is read and written O(1) times; the 17-tap windows are
// served by L1/L2.
#include <algorithm>
#include <cfloat>
#include <cstdlib>
#include <cmath>

#include "pxsom_common.h"

namespace {

#pragma clang fp contract(off)

constexpr int kMaxRadius = 64;

struct Taps {
    double w[kMaxRadius + 1];  // w[0] centre, w[d] weight at distance d (symmetric kernel)
    int radius;
};

// scipy NI_EXTEND_REFLECT: (d c b a | a b c d | d c b a)
__device__ __forceinline__ int reflect_idx(int i, int len)
{
    /* position i of the infinitely reflected line, for any i (the image may be shorter than the kernel radius:
     * scipy's NI_ExtendLine keeps reflecting, period 2 * len) */
    if (len == 1) return 0;
    const int sz2 = 2 * len;
    int m = i % sz2;
    if (m < 0) m += sz2;
    return m < len ? m : sz2 - 1 - m;
}

// One pass of scipy's correlate1d, symmetric-kernel branch:
//   tmp = in[0]*w[0];  for d = r .. 1:  tmp += (in[-d] + in[+d]) * w[d]
// AXIS 0: along image rows (stride W*C), AXIS 1: along image columns (stride C).
// R32: the image holds float32 values (widened): scipy then computes each line in binary64 but stores the pass's
// result as float32, so each pass rounds its output to float32.
//
// Generic form (any radius, tiny images): thread <-> one output element, every tap a coalesced read served by L1 / L2.
template <int AXIS, bool R32>
__global__ __launch_bounds__(256) void blur_pass_kernel(const double *__restrict__ in, double *__restrict__ out,
                                                        int H, int W, int C, Taps taps)
{
    const int64_t total = (int64_t)H * W * C;
    const int64_t wc = (int64_t)W * C;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int y = (int)(e / wc);
        const int64_t rem = e - (int64_t)y * wc;
        const int xcol = (int)(rem / C);
        const int pos = AXIS == 0 ? y : xcol, len = AXIS == 0 ? H : W;
        const int64_t stride = AXIS == 0 ? wc : C;
        const int64_t base = e - (int64_t)pos * stride;
        double tmp = in[e] * taps.w[0];
        const bool interior = pos >= taps.radius && pos + taps.radius < len;
        for (int d = taps.radius; d >= 1; d--) {
            const int lo = interior ? pos - d : reflect_idx(pos - d, len);
            const int hi = interior ? pos + d : reflect_idx(pos + d, len);
            tmp += (in[base + (int64_t)lo * stride] + in[base + (int64_t)hi * stride]) * taps.w[d];
        }
        out[e] = R32 ? (double)(float)tmp : tmp;
    }
}

// Tiles of the two fast forms are dealt to the XCDs in contiguous runs: workgroup b runs on XCD b % 8 (round robin),
// and every XCD has an L2 of its own -- neighbouring tiles share their halo, so they should meet in the same L2
// (the generic form above fetched 1.7x / 2.4x the image from HBM per pass: profiles/r03/preprocess.txt).
__device__ __forceinline__ int64_t xcd_contiguous(int64_t b, int64_t nb)
{
    constexpr int kXcds = 8;
    const int64_t per = (nb + kXcds - 1) / kXcds;
    return (b % kXcds) * per + b / kXcds;     // may be >= nb: the caller skips those
}

// AXIS 0, radius R (the pipeline's sigma = 2: R = 8): thread <-> one image column element (x, c), walking down a strip
// of rows with the last 2R+1 values of its column in REGISTERS -- every input is read once per strip (coalesced across
// the threads), every output is formed from registers with the reference's operation order.  Strip height TY: the
// halo of 2R rows is re-read by the strip below ((TY + 2R) / TY = 1.25x at TY = 64).
template <int R, bool R32>
__global__ __launch_bounds__(256) void blur_rows_window_kernel(const double *__restrict__ in, double *__restrict__ out,
                                                               int H, int64_t wc, Taps taps, int TY, int64_t ncb, int64_t nblocks)
{
    constexpr int NW = 2 * R + 1;
    const int64_t b = xcd_contiguous(blockIdx.x, nblocks);
    if (b >= nblocks) return;
    // strips of one column block are consecutive: their shared halo rows stay in one L2
    const int64_t cb = b / ((H + TY - 1) / TY), strip = b - cb * ((H + TY - 1) / TY);
    (void)ncb;
    const int64_t col = cb * 256 + threadIdx.x;
    if (col >= wc) return;
    const int y0 = (int)strip * TY, y1 = min(y0 + TY, H);
    double win[NW];   // win[i] = in[reflect(y - R + i)] for the output row y being formed
#pragma unroll
    for (int i = 0; i < NW; i++) win[i] = in[(int64_t)reflect_idx(y0 - R + i, H) * wc + col];
    double w[R + 1];
#pragma unroll
    for (int d = 0; d <= R; d++) w[d] = taps.w[d];
    for (int y = y0; y < y1; y += NW) {          // NW outputs per trip: the window rotates through the registers
#pragma unroll
        for (int u = 0; u < NW; u++) {
            if (y + u < y1) {
                // the window starts at register u (rotation by u): centre = win[(u + R) % NW]
                double tmp = win[(u + R) % NW] * w[0];
#pragma unroll
                for (int d = R; d >= 1; d--) tmp += (win[(u + R - d) % NW] + win[(u + R + d) % NW]) * w[d];
                out[(int64_t)(y + u) * wc + col] = R32 ? (double)(float)tmp : tmp;
                // the oldest value (register u) makes room for row y + u + R + 1
                win[u] = in[(int64_t)reflect_idx(y + u + R + 1, H) * wc + col];
            }
        }
    }
}

// AXIS 1, radius R: a workgroup stages a run of TX consecutive elements of one image row plus R pixels (R*C elements)
// of halo on either side in LDS -- reflection resolved while staging -- and forms TX outputs from it: every tap is an
// LDS read at stride C elements from its neighbour's (conflict-free: consecutive threads read consecutive words).
template <int R, bool R32>
__global__ __launch_bounds__(256) void blur_cols_lds_kernel(const double *__restrict__ in, double *__restrict__ out,
                                                            int H, int W, int C, Taps taps, int TX, int64_t nblocks)
{
    extern __shared__ double blur_tile[];     // [TX + 2 R C]
    const int64_t b = xcd_contiguous(blockIdx.x, nblocks);
    if (b >= nblocks) return;
    const int64_t wc = (int64_t)W * C;
    const int tiles_per_row = (int)((wc + TX - 1) / TX);
    const int y = (int)(b / tiles_per_row);
    const int64_t e0 = (b - (int64_t)y * tiles_per_row) * TX;
    const int halo = R * C, span = TX + 2 * halo;
    const double *row = in + (int64_t)y * wc;
    for (int i = threadIdx.x; i < span; i += 256) {
        const int64_t e = e0 - halo + i;                         // element of the row's infinite reflected extension
        // pixel index floor(e / C) reflected, channel kept
        int64_t px = e >= 0 ? e / C : -((-e + C - 1) / C);
        const int ch = (int)(e - px * C);
        blur_tile[i] = px >= W + R ? 0.0 : row[(int64_t)reflect_idx((int)px, W) * C + ch];   // (past the tile's last real output: unused)
    }
    __syncthreads();
    double w[R + 1];
#pragma unroll
    for (int d = 0; d <= R; d++) w[d] = taps.w[d];
    for (int i = threadIdx.x; i < TX && e0 + i < wc; i += 256) {
        const double *c0 = blur_tile + halo + i;
        double tmp = c0[0] * w[0];
#pragma unroll
        for (int d = R; d >= 1; d--) tmp += (c0[-d * C] + c0[d * C]) * w[d];
        out[(int64_t)y * wc + e0 + i] = R32 ? (double)(float)tmp : tmp;
    }
}

// ---- row filter + normalise + compaction ---------------------------------------------------------
// pass A: keep flag per row and per-workgroup kept counts; pass B: exclusive scan of the counts (one
// workgroup); pass C: rows rewritten as x / rowsum at their compacted position, in pixel order.
// f32: the matrix holds float32 values (widened) and pandas works on a float32 frame: sum and division in
// binary32.
__device__ __forceinline__ bool row_keep(const double *__restrict__ row, int c, double thresh, double &s, int f32)
{
    bool any = false;
    if (f32) {
        float acc = 0.f;
        for (int j = 0; j < c; j++) {
            const float v = (float)row[j];
            acc += v;
            any |= (v != 0.f);
        }
        s = (double)acc;
    } else {
        double acc = 0.0;  // left-to-right, as pandas' DataFrame.sum(axis=1) adds the columns
        for (int j = 0; j < c; j++) {
            const double v = row[j];
            acc += v;
            any |= (v != 0.0);
        }
        s = acc;
    }
    return (s > thresh) && any;
}

// A workgroup owns 256 consecutive rows.  The rows are staged in LDS with coalesced loads (element e of the block's
// 256 * c elements by thread e % 256), each thread then sums ITS row from LDS left to right -- the reference's order --
// at a padded stride (c | 1 doubles: a thread-per-row walk over global memory costs a cache line per lane and load).
constexpr int kRowBlock = 256;
__host__ __device__ __forceinline__ int row_pad(int c) { return c | 1; }

__device__ __forceinline__ void stage_rows(const double *__restrict__ x, int64_t row0, int64_t n, int c, double *tile)
{
    const int64_t e_end = min((int64_t)kRowBlock, n - row0) * c;
    const double *src = x + row0 * c;
    const int cp = row_pad(c);
    int r = (int)threadIdx.x / c, j = (int)threadIdx.x - r * c;       // element e <-> (row r, channel j), no division per element
    const int dr = kRowBlock / c, dj = kRowBlock % c;
    for (int64_t e = threadIdx.x; e < e_end; e += kRowBlock) {
        tile[r * cp + j] = src[e];
        r += dr;
        j += dj;
        if (j >= c) {
            j -= c;
            r++;
        }
    }
}

__global__ __launch_bounds__(256) void rowfilter_count_kernel(const double *__restrict__ x, int64_t n, int c,
                                                              double thresh, unsigned *__restrict__ block_counts,
                                                              int f32)
{
    extern __shared__ double rf_tile[];
    __shared__ unsigned s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    const int64_t row0 = (int64_t)blockIdx.x * kRowBlock;
    stage_rows(x, row0, n, c, rf_tile);
    __syncthreads();
    const int64_t row = row0 + threadIdx.x;
    bool keep = false;
    if (row < n) {
        double s;
        keep = row_keep(rf_tile + (size_t)threadIdx.x * row_pad(c), c, thresh, s, f32);
    }
    const unsigned long long m = __ballot(keep);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&s_cnt, (unsigned)__popcll(m));
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = s_cnt;
}

__global__ __launch_bounds__(1024) void block_scan_kernel(unsigned *__restrict__ block_counts, int64_t nblocks,
                                                          int64_t *__restrict__ total_out)
{
    // exclusive scan in place (int64 running offset kept in shared memory between 1024-wide sweeps)
    __shared__ unsigned long long s_part[1024];
    __shared__ unsigned long long s_base;
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    for (int64_t b0 = 0; b0 < nblocks; b0 += 1024) {
        const int64_t i = b0 + threadIdx.x;
        const unsigned v = i < nblocks ? block_counts[i] : 0u;
        s_part[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const unsigned long long t = threadIdx.x >= off ? s_part[threadIdx.x - off] : 0ull;
            __syncthreads();
            s_part[threadIdx.x] += t;
            __syncthreads();
        }
        const unsigned long long incl = s_part[threadIdx.x], base = s_base;
        if (i < nblocks) block_counts[i] = (unsigned)(base + incl - v);  // < 2^32 rows per call
        __syncthreads();
        if (threadIdx.x == 1023) s_base = base + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total_out = (int64_t)s_base;
}

__global__ __launch_bounds__(256) void rowfilter_write_kernel(const double *__restrict__ x, int64_t n, int c,
                                                              double thresh, const unsigned *__restrict__ block_off,
                                                              double *__restrict__ out_rows,
                                                              int64_t *__restrict__ out_index, int f32)
{
    extern __shared__ double rf_tile[];                    // [256][c | 1] rows, then [256] row sums, then [256] u16 kept rows
    __shared__ unsigned s_wave[4];
    const int cp = row_pad(c);
    double *sums = rf_tile + (size_t)kRowBlock * cp;
    unsigned short *kept = reinterpret_cast<unsigned short *>(sums + kRowBlock);
    const int64_t row0 = (int64_t)blockIdx.x * kRowBlock;
    stage_rows(x, row0, n, c, rf_tile);
    __syncthreads();
    const int64_t row = row0 + threadIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    bool keep = false;
    double s = 0.0;
    if (row < n) keep = row_keep(rf_tile + (size_t)threadIdx.x * cp, c, thresh, s, f32);
    const unsigned long long m = __ballot(keep);
    if (lane == 0) s_wave[wv] = (unsigned)__popcll(m);
    __syncthreads();
    unsigned local = 0;
    for (int i = 0; i < wv; i++) local += s_wave[i];
    const unsigned nkept = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    const int64_t dst0 = block_off[blockIdx.x];
    if (keep) {
        const unsigned slot = local + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
        kept[slot] = (unsigned short)threadIdx.x;
        sums[slot] = s;
        out_index[dst0 + slot] = row;
    }
    __syncthreads();
    // the kept rows are consecutive in the output: element e of the block's nkept * c outputs by thread e % 256
    double *dst = out_rows + dst0 * c;
    int r = (int)threadIdx.x / c, j = (int)threadIdx.x - r * c;
    const int dr = kRowBlock / c, dj = kRowBlock % c;
    for (int e = threadIdx.x; e < (int)nkept * c; e += kRowBlock) {
        const double v = rf_tile[(size_t)kept[r] * cp + j], sr = sums[r];
        dst[e] = f32 ? (double)((float)v / (float)sr) : v / sr;
        r += dr;
        j += dj;
        if (j >= c) {
            j -= c;
            r++;
        }
    }
}

// Wide rows (c > 72: 256 staged rows would not fit the LDS): thread <-> row straight from global memory.
__global__ __launch_bounds__(256) void rowfilter_count_direct_kernel(const double *__restrict__ x, int64_t n, int c,
                                                                     double thresh, unsigned *__restrict__ block_counts,
                                                                     int f32)
{
    __shared__ unsigned s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    bool keep = false;
    if (row < n) {
        double s;
        keep = row_keep(x + row * c, c, thresh, s, f32);
    }
    const unsigned long long m = __ballot(keep);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&s_cnt, (unsigned)__popcll(m));
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = s_cnt;
}

__global__ __launch_bounds__(256) void rowfilter_write_direct_kernel(const double *__restrict__ x, int64_t n, int c,
                                                                     double thresh, const unsigned *__restrict__ block_off,
                                                                     double *__restrict__ out_rows,
                                                                     int64_t *__restrict__ out_index, int f32)
{
    __shared__ unsigned s_wave[4];
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    bool keep = false;
    double s = 0.0;
    if (row < n) keep = row_keep(x + row * c, c, thresh, s, f32);
    const unsigned long long m = __ballot(keep);
    if (lane == 0) s_wave[wv] = (unsigned)__popcll(m);
    __syncthreads();
    unsigned off = block_off[blockIdx.x];
    for (int i = 0; i < wv; i++) off += s_wave[i];
    if (keep) {
        const int64_t dst = (int64_t)off + __popcll(m & ((1ull << lane) - 1ull));
        const double *src = x + row * c;
        double *d = out_rows + dst * c;
        if (f32)
            for (int j = 0; j < c; j++) d[j] = (double)((float)src[j] / (float)s);
        else
            for (int j = 0; j < c; j++) d[j] = src[j] / s;
        out_index[dst] = row;
    }
}

__global__ __launch_bounds__(256) void normalize_columns_kernel(const double *__restrict__ x, int64_t n, int c,
                                                                int64_t ldx, const double *__restrict__ norm,
                                                                double *__restrict__ out, int64_t ldo)
{
    const int64_t total = n * c;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t row = e / c;
        const int j = (int)(e - row * c);
        out[row * ldo + j] = x[row * ldx + j] / norm[j];
    }
}

// ---- exact quantile of the non-zero values of each column (MSB-first radix select on binary64) ----
// key(v): order-preserving 64-bit integer image of a double.
__device__ __forceinline__ unsigned long long f64_key(double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_f64(unsigned long long k)
{
    const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}
// keep_mode 0: != 0 and not NaN (pandas replace(0, nan)); 1: > 0 (img[img > 0]); 2: everything but NaN
__device__ __forceinline__ bool q_keep(double v, int keep_mode)
{
    return keep_mode == 0 ? (v != 0.0 && v == v) : (keep_mode == 1 ? v > 0.0 : v == v);
}

struct QState {              // per column, in the workspace
    unsigned long long prefix;   // key bits fixed so far (high bits)
    unsigned long long m;        // number of kept values
    unsigned long long rank;     // rank still to find inside the current prefix bucket
    unsigned long long lo_key, hi_key;
    unsigned long long count_le; // kept values with key <= lo_key
};

// pass (shift = 56, 48, ..., 0): histogram of byte (key >> shift) over kept values whose higher bits equal
// prefix.  hist [c][256] u64 in the workspace (zeroed before every pass).
// ONE sweep over the matrix serves a chunk of up to kQCols columns: consecutive threads read consecutive elements of a
// row (a kernel per column read 8 bytes of every cache line: 12x the matrix from HBM per pass, profiles/r03), every
// column has its own 256-bin histogram in LDS.
constexpr int kQCols = 48, kQRows = 1024;
template <typename T>
__global__ __launch_bounds__(256) void q_hist_kernel(const T *__restrict__ x, int64_t n, int c, int64_t ldx,
                                                     int keep_mode, int shift, const QState *__restrict__ st,
                                                     unsigned long long *__restrict__ hist, int first_pass, int col0, int ncols)
{
    __shared__ unsigned s_h[kQCols * 256];
    __shared__ unsigned long long s_prefix[kQCols];
    (void)c;
    for (int i = threadIdx.x; i < ncols * 256; i += 256) s_h[i] = 0;
    if ((int)threadIdx.x < ncols) s_prefix[threadIdx.x] = first_pass ? 0ull : st[col0 + threadIdx.x].prefix;
    __syncthreads();
    const unsigned long long himask = shift >= 56 ? 0ull : (~0ull << (shift + 8));
    const int dr = 256 / ncols, dj = 256 % ncols;
    for (int64_t row0 = (int64_t)blockIdx.x * kQRows; row0 < n; row0 += (int64_t)gridDim.x * kQRows) {
        const int e_end = (int)min((int64_t)kQRows, n - row0) * ncols;
        int r = (int)threadIdx.x / ncols, j = (int)threadIdx.x - r * ncols;
        for (int e = threadIdx.x; e < e_end; e += 4 * 256) {     // four loads in flight per thread
            double v[4];
            int jj[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                jj[u] = j;
                v[u] = e + u * 256 < e_end ? (double)x[(row0 + r) * ldx + col0 + j] : 0.0;   // binary32 values are exact in binary64
                r += dr;
                j += dj;
                if (j >= ncols) {
                    j -= ncols;
                    r++;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (e + u * 256 < e_end && q_keep(v[u], keep_mode)) {
                    const unsigned long long k = f64_key(v[u]);
                    if ((k & himask) == (s_prefix[jj[u]] & himask)) atomicAdd(&s_h[jj[u] * 256 + ((unsigned)(k >> shift) & 255u)], 1u);
                }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ncols * 256; i += 256)
        if (s_h[i]) atomicAdd(&hist[(size_t)col0 * 256 + i], (unsigned long long)s_h[i]);
}

// one thread per column: locate the bucket holding `rank`, narrow the prefix.  On the first pass also
// derive m and the target rank lo = floor(q*(m-1)).
// arith32: the virtual index is formed in binary32 -- what numpy does for a float32 array
// (q is cast to the array's dtype, then (n-1)*q, floor and the fraction are all float32).
// fill_low: the rows are binary32 values, whose binary64 images end in 29 zero bits -- key bits below `fill_low` are all
// zeros (positive values) or all ones (negative ones: the key is the complement) for every candidate, so the passes over
// them are not run: this pass completes the prefix.
__global__ __launch_bounds__(64) void q_select_kernel(QState *st, unsigned long long *hist, int c, int shift, double q,
                                                      int first_pass, int arith32, int fill_low)
{
    // one wave per column: lane l owns bins 4 l .. 4 l + 3; an inclusive scan over the lanes' sums finds the bucket
    const int col = blockIdx.x, lane = threadIdx.x;
    if (col >= c) return;
    unsigned long long *h = hist + (size_t)col * 256;
    unsigned long long b4[4], mine = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        b4[i] = h[4 * lane + i];
        mine += b4[i];
        h[4 * lane + i] = 0;   // ready for the next pass
    }
    unsigned long long incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    const unsigned long long total = __shfl(incl, 63);
    QState s = st[col];
    if (first_pass) {
        s.m = total;
        s.prefix = 0;
        const double vi = arith32 ? (double)((float)(total > 0 ? total - 1 : 0) * (float)q) : q * (double)(total > 0 ? total - 1 : 0);
        unsigned long long lo = (unsigned long long)floor(vi);
        if (total > 0 && lo > total - 1) lo = total - 1;
        s.rank = lo;
    }
    if (s.m > 0) {
        // the first bin b with (sum of bins <= b) > rank; rank < m always holds, the clamp to bin 255 mirrors the serial form
        const unsigned long long excl = incl - mine;
        int bin = -1;
        unsigned long long before = 0, acc = excl;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (bin < 0 && acc + b4[i] > s.rank) {
                bin = 4 * lane + i;
                before = acc;
            }
            acc += b4[i];
        }
        const unsigned long long has = __ballot(bin >= 0);
        int src = has ? __builtin_ctzll(has) : 63;
        int wbin = __shfl(bin, src);
        unsigned long long wbefore = __shfl(before, src);
        if (!has) {   // cannot happen while rank < m: keep the serial form's behaviour (last bin, everything before it)
            wbin = 255;
            wbefore = total - __shfl(b4[3], 63);
        }
        s.rank -= wbefore;
        s.prefix |= (unsigned long long)wbin << shift;
        if (fill_low > 0 && !(s.prefix >> 63)) s.prefix |= (1ull << fill_low) - 1ull;
    }
    if (lane == 0) st[col] = s;
}

// after the last pass prefix == key of the order statistic lo.  One more sweep (same shape as q_hist_kernel): count
// keys <= lo_key and find the smallest key above it (the next order statistic unless lo_key is repeated).
template <typename T>
__global__ __launch_bounds__(256) void q_next_kernel(const T *__restrict__ x, int64_t n, int c, int64_t ldx,
                                                     int keep_mode, QState *st, int col0, int ncols)
{
    __shared__ unsigned long long s_min[kQCols], s_lo[kQCols];
    __shared__ unsigned s_cnt[kQCols];
    (void)c;
    if ((int)threadIdx.x < ncols) {
        s_min[threadIdx.x] = ~0ull;
        s_cnt[threadIdx.x] = 0;
        s_lo[threadIdx.x] = st[col0 + threadIdx.x].prefix;
    }
    __syncthreads();
    const int dr = 256 / ncols, dj = 256 % ncols;
    for (int64_t row0 = (int64_t)blockIdx.x * kQRows; row0 < n; row0 += (int64_t)gridDim.x * kQRows) {
        const int e_end = (int)min((int64_t)kQRows, n - row0) * ncols;
        int r = (int)threadIdx.x / ncols, j = (int)threadIdx.x - r * ncols;
        for (int e = threadIdx.x; e < e_end; e += 256) {
            const double v = (double)x[(row0 + r) * ldx + col0 + j];
            if (q_keep(v, keep_mode)) {
                const unsigned long long k = f64_key(v);
                if (k <= s_lo[j]) atomicAdd(&s_cnt[j], 1u);
                else if (k < s_min[j]) atomicMin(&s_min[j], k);     // (the plain read only skips hopeless atomics)
            }
            r += dr;
            j += dj;
            if (j >= ncols) {
                j -= ncols;
                r++;
            }
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < ncols) {
        atomicMin(&st[col0 + threadIdx.x].hi_key, s_min[threadIdx.x]);
        atomicAdd(&st[col0 + threadIdx.x].count_le, (unsigned long long)s_cnt[threadIdx.x]);
    }
}

// numpy's linear interpolation (_lerp): a + (b-a)*g for g < 0.5, b - (b-a)*(1-g) otherwise.
__global__ void q_finish_kernel(const QState *st, int c, double q, double *out, int arith32)
{
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= c) return;
    const QState s = st[col];
    if (s.m == 0) {
        out[col] = __longlong_as_double(0x7ff8000000000000ll);  // NaN, like pandas for an all-zero column
        return;
    }
    if (arith32) {  // numpy on a float32 array: index, fraction and interpolation in binary32
        const float vi = (float)(s.m - 1) * (float)q;
        unsigned long long lo = (unsigned long long)floorf(vi);
        if (lo > s.m - 1) lo = s.m - 1;
        const unsigned long long hi = lo + 1 > s.m - 1 ? s.m - 1 : lo + 1;
        const float g = vi - (float)lo;
        const float a = (float)key_f64(s.prefix);
        const float b = (hi == lo || s.count_le > hi) ? a : (float)key_f64(s.hi_key);
        const float diff = b - a;
        out[col] = (double)(g >= 0.5f ? b - diff * (1.0f - g) : a + diff * g);
        return;
    }
    const double vi = q * (double)(s.m - 1);
    unsigned long long lo = (unsigned long long)floor(vi);
    if (lo > s.m - 1) lo = s.m - 1;
    const unsigned long long hi = lo + 1 > s.m - 1 ? s.m - 1 : lo + 1;
    const double g = vi - (double)lo;
    const double a = key_f64(s.prefix);
    // order statistic hi: still lo's value if that value occupies ranks beyond lo
    const double b = (hi == lo || s.count_le > hi) ? a : key_f64(s.hi_key);
    const double diff = b - a;
    out[col] = g >= 0.5 ? b - diff * (1.0 - g) : a + diff * g;
}

__global__ void q_init_kernel(QState *st, int c)
{
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= c) return;
    QState s;
    s.prefix = 0;
    s.m = 0;
    s.rank = 0;
    s.lo_key = 0;
    s.hi_key = ~0ull;
    s.count_le = 0;
    st[col] = s;
}

#pragma clang fp contract(fast)

}  // namespace

PXSOM_EXPORT int pxsom_gaussian_blur_hwc(double *img_dev, double *tmp_dev, int h, int w, int c,
                                         const double *weights_host, int radius, int f32_semantics, void *stream)
{
    if (!img_dev || !tmp_dev || !weights_host || h < 1 || w < 1 || c < 1)
        return pxsom::fail(PXSOM_ERR_INVALID_ARG, "pxsom_gaussian_blur_hwc: bad arguments");
    if (radius < 0 || radius > kMaxRadius)
        return pxsom::fail(PXSOM_ERR_UNSUPPORTED, "pxsom_gaussian_blur_hwc: radius %d outside [0, %d]", radius, kMaxRadius);
    Taps taps;
    taps.radius = radius;
    for (int d = 0; d <= radius; d++) taps.w[d] = weights_host[radius + d];  // symmetric: w[r+d] == w[r-d]
    for (int d = radius + 1; d <= kMaxRadius; d++) taps.w[d] = 0.0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t total = (int64_t)h * w * c;
    const int64_t wc = (int64_t)w * c;
    const bool generic = (f32_semantics & PXSOM_BLUR_GENERIC_FORM) != 0;   // tests compare the two forms
    f32_semantics &= 1;
    // the pipeline's kernel (sigma = 2 -> radius 8) on images of at least a strip: register window down the rows, LDS
    // tile along the columns; anything else (other radii, images shorter than the window) takes the generic form
    if (radius == 8 && !generic && h >= 17 && w >= 9 && c <= 256) {   // (tile + halo of the column pass: <= 41 KB of LDS)
        constexpr int R = 8;
        const int TY = 64;
        const int64_t ncb = (wc + 255) / 256, strips = (h + TY - 1) / TY, nb0 = ncb * strips;
        const int64_t g0 = (nb0 + 7) / 8 * 8;
        const int TX = 1024;
        const int64_t nb1 = (int64_t)h * ((wc + TX - 1) / TX), g1 = (nb1 + 7) / 8 * 8;
        const size_t lds1 = (size_t)(TX + 2 * R * c) * sizeof(double);
        if (f32_semantics) {
            hipLaunchKernelGGL((blur_rows_window_kernel<R, true>), dim3((unsigned)g0), dim3(256), 0, st, img_dev, tmp_dev, h, wc, taps, TY, ncb, nb0);
            hipLaunchKernelGGL((blur_cols_lds_kernel<R, true>), dim3((unsigned)g1), dim3(256), lds1, st, tmp_dev, img_dev, h, w, c, taps, TX, nb1);
        } else {
            hipLaunchKernelGGL((blur_rows_window_kernel<R, false>), dim3((unsigned)g0), dim3(256), 0, st, img_dev, tmp_dev, h, wc, taps, TY, ncb, nb0);
            hipLaunchKernelGGL((blur_cols_lds_kernel<R, false>), dim3((unsigned)g1), dim3(256), lds1, st, tmp_dev, img_dev, h, w, c, taps, TX, nb1);
        }
        PXSOM_LAUNCH_CHECK("blur_rows_window_kernel / blur_cols_lds_kernel");
        return PXSOM_OK;
    }
    const int grid = (int)std::min<int64_t>((total + 255) / 256, (int64_t)pxsom::device_cu_count() * 16);
    if (f32_semantics) {
        hipLaunchKernelGGL((blur_pass_kernel<0, true>), dim3(grid), dim3(256), 0, st, img_dev, tmp_dev, h, w, c, taps);
        hipLaunchKernelGGL((blur_pass_kernel<1, true>), dim3(grid), dim3(256), 0, st, tmp_dev, img_dev, h, w, c, taps);
    } else {
        hipLaunchKernelGGL((blur_pass_kernel<0, false>), dim3(grid), dim3(256), 0, st, img_dev, tmp_dev, h, w, c, taps);
        hipLaunchKernelGGL((blur_pass_kernel<1, false>), dim3(grid), dim3(256), 0, st, tmp_dev, img_dev, h, w, c, taps);
    }
    PXSOM_LAUNCH_CHECK("blur_pass_kernel");
    return PXSOM_OK;
}

PXSOM_EXPORT size_t pxsom_rownorm_workspace_bytes(int64_t n)
{
    if (n < 0) return 0;
    return pxsom::align_up((size_t)((n + 255) / 256 + 1) * sizeof(unsigned), 256);
}

PXSOM_EXPORT int pxsom_rowsum_filter_normalize(const double *x_dev, int64_t n, int c, double thresh,
                                               double *out_rows_dev, int64_t *out_index_dev, int64_t *out_count_dev,
                                               void *workspace_dev, size_t workspace_bytes, int f32_semantics,
                                               void *stream)
{
    if (n < 0 || n > 0x7fffffffLL || c < 1 || !out_count_dev || (n > 0 && (!x_dev || !out_rows_dev || !out_index_dev)))
        return pxsom::fail(PXSOM_ERR_INVALID_ARG, "pxsom_rowsum_filter_normalize: bad arguments");

    if (!workspace_dev || workspace_bytes < pxsom_rownorm_workspace_bytes(n))
        return pxsom::fail(PXSOM_ERR_WORKSPACE, "pxsom_rowsum_filter_normalize: workspace too small");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (n == 0) {
        PXSOM_HIP_TRY(hipMemsetAsync(out_count_dev, 0, sizeof(int64_t), st));
        return PXSOM_OK;
    }
    unsigned *counts = reinterpret_cast<unsigned *>(workspace_dev);
    const int64_t nblocks = (n + 255) / 256;
    const size_t tile_bytes = (size_t)kRowBlock * row_pad(c) * sizeof(double);
    const size_t write_bytes = tile_bytes + kRowBlock * sizeof(double) + kRowBlock * sizeof(unsigned short);
    const bool staged = c <= 72;   // 256 rows x (c | 1) doubles of LDS per workgroup
    if (staged && write_bytes > 64 * 1024) {   // c > 30: beyond the default dynamic LDS limit
        static pxsom::PerDevice<size_t> raised;
        size_t &have = raised.here();
        if (have < write_bytes) {
            PXSOM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(rowfilter_count_kernel),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)write_bytes));
            PXSOM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(rowfilter_write_kernel),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)write_bytes));
            have = write_bytes;
        }
    }
    if (staged)
        hipLaunchKernelGGL(rowfilter_count_kernel, dim3((unsigned)nblocks), dim3(256), tile_bytes, st, x_dev, n, c, thresh, counts,
                           f32_semantics);
    else
        hipLaunchKernelGGL(rowfilter_count_direct_kernel, dim3((unsigned)nblocks), dim3(256), 0, st, x_dev, n, c, thresh, counts,
                           f32_semantics);
    PXSOM_LAUNCH_CHECK("rowfilter_count_kernel");
    hipLaunchKernelGGL(block_scan_kernel, dim3(1), dim3(1024), 0, st, counts, nblocks, out_count_dev);
    PXSOM_LAUNCH_CHECK("block_scan_kernel");
    if (staged)
        hipLaunchKernelGGL(rowfilter_write_kernel, dim3((unsigned)nblocks), dim3(256), write_bytes, st, x_dev, n, c, thresh, counts,
                           out_rows_dev, out_index_dev, f32_semantics);
    else
        hipLaunchKernelGGL(rowfilter_write_direct_kernel, dim3((unsigned)nblocks), dim3(256), 0, st, x_dev, n, c, thresh, counts,
                           out_rows_dev, out_index_dev, f32_semantics);
    PXSOM_LAUNCH_CHECK("rowfilter_write_kernel");
    return PXSOM_OK;
}

PXSOM_EXPORT int pxsom_normalize_columns(const double *x_dev, int64_t n, int c, int64_t ldx, const double *norm_dev,
                                         double *out_dev, int64_t ldo, void *stream)
{
    if (n < 0 || c < 1 || ldx < c || ldo < c || !norm_dev || (n > 0 && (!x_dev || !out_dev)))
        return pxsom::fail(PXSOM_ERR_INVALID_ARG, "pxsom_normalize_columns: bad arguments");
    if (n == 0) return PXSOM_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int grid = (int)std::min<int64_t>((n * c + 255) / 256, (int64_t)pxsom::device_cu_count() * 16);
    hipLaunchKernelGGL(normalize_columns_kernel, dim3(grid), dim3(256), 0, st, x_dev, n, c, ldx, norm_dev, out_dev, ldo);
    PXSOM_LAUNCH_CHECK("normalize_columns_kernel");
    return PXSOM_OK;
}

PXSOM_EXPORT size_t pxsom_quantile_workspace_bytes(int64_t n, int c)
{
    if (n < 0 || c < 1) return 0;
    (void)n;
    return pxsom::align_up((size_t)c * sizeof(QState), 256) + (size_t)c * 256 * sizeof(unsigned long long);
}

namespace {
template <typename T>
int quantile_typed(const char *fn, const T *x_dev, int64_t n, int c, int64_t ldx, double q, int keep_mode,
                   double *out_dev, void *workspace_dev, size_t workspace_bytes, int arith32, hipStream_t st)
{
    if (n < 0 || c < 1 || c > 65535 || ldx < c || !(q >= 0.0 && q <= 1.0) || !out_dev || (n > 0 && !x_dev) ||
        keep_mode < 0 || keep_mode > 2)
        return pxsom::fail(PXSOM_ERR_INVALID_ARG, "%s: bad arguments", fn);
    if (!workspace_dev || workspace_bytes < pxsom_quantile_workspace_bytes(n, c))
        return pxsom::fail(PXSOM_ERR_WORKSPACE, "%s: workspace too small", fn);
    QState *qs = reinterpret_cast<QState *>(workspace_dev);
    unsigned long long *hist =
        reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(workspace_dev) + pxsom::align_up((size_t)c * sizeof(QState), 256));
    const int cgrid = (c + 63) / 64;
    hipLaunchKernelGGL(q_init_kernel, dim3(cgrid), dim3(64), 0, st, qs, c);
    PXSOM_HIP_TRY(hipMemsetAsync(hist, 0, (size_t)c * 256 * sizeof(unsigned long long), st));
    int rgrid = (int)std::min<int64_t>((n + kQRows - 1) / kQRows, (int64_t)pxsom::device_cu_count() * 4);
    if (rgrid < 1) rgrid = 1;
    // binary32 rows: key bits 0 .. 23 follow from the sign (three sweeps fewer)
    const int last_shift = sizeof(T) == 4 ? 24 : 0;
    for (int shift = 56, first = 1; shift >= last_shift; shift -= 8, first = 0) {
        for (int col0 = 0; col0 < c; col0 += kQCols)
            hipLaunchKernelGGL(q_hist_kernel<T>, dim3(rgrid), dim3(256), 0, st, x_dev, n, c, ldx, keep_mode, shift, qs, hist, first,
                               col0, std::min(kQCols, c - col0));
        hipLaunchKernelGGL(q_select_kernel, dim3(c), dim3(64), 0, st, qs, hist, c, shift, q, first, arith32,
                           (shift == last_shift && last_shift > 0) ? last_shift : 0);
    }
    for (int col0 = 0; col0 < c; col0 += kQCols)
        hipLaunchKernelGGL(q_next_kernel<T>, dim3(rgrid), dim3(256), 0, st, x_dev, n, c, ldx, keep_mode, qs, col0,
                           std::min(kQCols, c - col0));
    hipLaunchKernelGGL(q_finish_kernel, dim3(cgrid), dim3(64), 0, st, qs, c, q, out_dev, arith32);
    PXSOM_LAUNCH_CHECK(fn);
    return PXSOM_OK;
}
}  // namespace

PXSOM_EXPORT int pxsom_quantile_nonzero(const double *x_dev, int64_t n, int c, int64_t ldx, double q, int keep_mode,
                                        double *out_dev, void *workspace_dev, size_t workspace_bytes, void *stream)
{
    return quantile_typed<double>("pxsom_quantile_nonzero", x_dev, n, c, ldx, q, keep_mode, out_dev, workspace_dev,
                                  workspace_bytes, 0, reinterpret_cast<hipStream_t>(stream));
}

// float32 columns with numpy's float32 arithmetic (np.quantile on a float32 image: the TIFF-side percentiles
// of calculate_channel_percentiles / calculate_pixel_intensity_percentile).  out_dev holds the float32
// results widened to binary64.
PXSOM_EXPORT int pxsom_quantile_f32(const float *x_dev, int64_t n, int c, int64_t ldx, double q, int keep_mode,
                                    double *out_dev, void *workspace_dev, size_t workspace_bytes, void *stream)
{
    return quantile_typed<float>("pxsom_quantile_f32", x_dev, n, c, ldx, q, keep_mode, out_dev, workspace_dev,
                                 workspace_bytes, 1, reinterpret_cast<hipStream_t>(stream));
}

// out[i] = sum_j img[i, j] / norm[j] in float32, added up the way numpy adds a contiguous float32 axis of at
// most 128 elements (np.sum(img / norm_vect, axis=-1), pixel_cluster_utils.py:96-101): fewer than 8 terms
// left to right; otherwise eight running sums over blocks of 8, combined ((0+1)+(2+3))+((4+5)+(6+7)), then the
// remaining terms one by one.
namespace {
#pragma clang fp contract(off)
template <typename F>
__global__ __launch_bounds__(256) void scaled_rowsum_kernel(const F *__restrict__ img, int64_t n, int c,
                                                            int64_t ldx, const F *__restrict__ norm,
                                                            F *__restrict__ out)
{
    for (int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x; row < n; row += (int64_t)gridDim.x * 256) {
        const F *p = img + row * ldx;
        F res;
        if (c < 8) {
            res = (F)0;
            for (int j = 0; j < c; j++) res += p[j] / norm[j];
        } else {
            F r[8];
#pragma unroll
            for (int j = 0; j < 8; j++) r[j] = p[j] / norm[j];
            int i = 8;
            for (; i < c - (c % 8); i += 8) {
#pragma unroll
                for (int j = 0; j < 8; j++) r[j] += p[i + j] / norm[i + j];
            }
            res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
            for (; i < c; i++) res += p[i] / norm[i];
        }
        out[row] = res;
    }
}
#pragma clang fp contract(fast)

template <typename F>
int scaled_rowsum(const char *fn, const F *img_dev, int64_t n, int c, int64_t ldx, const F *norm_dev, F *out_dev,
                  void *stream)
{
    if (n < 0 || c < 1 || c > 128 || ldx < c || !norm_dev || (n > 0 && (!img_dev || !out_dev)))
        return pxsom::fail(PXSOM_ERR_INVALID_ARG, "%s: bad arguments (c <= 128)", fn);
    if (n == 0) return PXSOM_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t grid = std::min<int64_t>((n + 255) / 256, (int64_t)pxsom::device_cu_count() * 16);
    hipLaunchKernelGGL(scaled_rowsum_kernel<F>, dim3((unsigned)grid), dim3(256), 0, st, img_dev, n, c, ldx, norm_dev,
                       out_dev);
    PXSOM_LAUNCH_CHECK("scaled_rowsum_kernel");
    return PXSOM_OK;
}
}  // namespace

PXSOM_EXPORT int pxsom_scaled_rowsum_f32(const float *img_dev, int64_t n, int c, int64_t ldx, const float *norm_dev,
                                         float *out_dev, void *stream)
{
    return scaled_rowsum<float>("pxsom_scaled_rowsum_f32", img_dev, n, c, ldx, norm_dev, out_dev, stream);
}

PXSOM_EXPORT int pxsom_scaled_rowsum_f64(const double *img_dev, int64_t n, int c, int64_t ldx, const double *norm_dev,
                                         double *out_dev, void *stream)
{
    return scaled_rowsum<double>("pxsom_scaled_rowsum_f64", img_dev, n, c, ldx, norm_dev, out_dev, stream);
}

// ------------------------------------------------------------------------------------------------
// pixel cluster mask (generate_pixel_cluster_mask): winner[pos] = highest row index listing that pixel
// (numpy's sequential fancy assignment keeps the last), then mask[pos] = lut[label[winner]]
// ------------------------------------------------------------------------------------------------
namespace {

__global__ __launch_bounds__(256) void mask_winner_kernel(const int64_t *__restrict__ row_index,
                                                          const int64_t *__restrict__ column_index,
                                                          const int64_t *__restrict__ labels, int64_t n,
                                                          const int32_t *__restrict__ lut, int64_t lut_size, int h,
                                                          int w, long long *winner, int32_t *status)
{
    int bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = row_index[i], c = column_index[i], lb = labels[i];
        const int64_t pos = r * w + c;  // numpy indexes the flattened image with this number
        if (pos < 0 || pos >= (int64_t)h * w) {
            bad |= PXSOM_MASK_BAD_PIXEL;
            continue;
        }
        if (lb < 0 || lb >= lut_size || lut[lb] == PXSOM_LUT_UNMAPPED) bad |= PXSOM_MASK_BAD_LABEL;
        atomicMax(&winner[pos], (long long)i);
    }
    if (bad) atomicOr(status, bad);
}

__global__ __launch_bounds__(256) void mask_resolve_kernel(const int64_t *__restrict__ labels,
                                                           const int32_t *__restrict__ lut, int64_t lut_size,
                                                           int64_t pixels, const long long *__restrict__ winner,
                                                           int16_t *mask)
{
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < pixels; p += (int64_t)gridDim.x * 256) {
        const long long i = winner[p];
        int32_t id = 0;
        if (i >= 0) {
            const int64_t lb = labels[i];
            id = (lb >= 0 && lb < lut_size) ? lut[lb] : 0;
        }
        mask[p] = (int16_t)id;
    }
}

}  // namespace

PXSOM_EXPORT size_t pxsom_cluster_mask_workspace_bytes(int h, int w)
{
    return h > 0 && w > 0 ? (size_t)h * (size_t)w * sizeof(long long) : 0;
}

PXSOM_EXPORT int pxsom_cluster_mask(const int64_t *row_index_dev, const int64_t *column_index_dev,
                                    const int64_t *labels_dev, int64_t n, const int32_t *lut_dev, int64_t lut_size,
                                    int h, int w, int16_t *mask_dev, int32_t *status_dev, void *workspace_dev,
                                    size_t workspace_bytes, void *stream)
{
    if (n < 0 || h < 1 || w < 1 || lut_size < 0)
        return pxsom::fail(PXSOM_ERR_INVALID_ARG, "pxsom_cluster_mask: bad sizes (n=%lld, %dx%d)", (long long)n, h, w);
    if (!mask_dev || !status_dev || !workspace_dev || (lut_size > 0 && !lut_dev) ||
        (n > 0 && (!row_index_dev || !column_index_dev || !labels_dev)))
        return pxsom::fail(PXSOM_ERR_INVALID_ARG, "pxsom_cluster_mask: null pointer");
    const size_t need = pxsom_cluster_mask_workspace_bytes(h, w);
    if (workspace_bytes < need)
        return pxsom::fail(PXSOM_ERR_INVALID_ARG, "pxsom_cluster_mask: workspace %zu < %zu bytes", workspace_bytes, need);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t pixels = (int64_t)h * w;
    PXSOM_HIP_TRY(hipMemsetAsync(workspace_dev, 0xFF, need, st));  // every winner = -1
    PXSOM_HIP_TRY(hipMemsetAsync(status_dev, 0, sizeof(int32_t), st));
    const int64_t cap = (int64_t)pxsom::device_cu_count() * 16;
    long long *winner = reinterpret_cast<long long *>(workspace_dev);
    if (n > 0) {
        hipLaunchKernelGGL(mask_winner_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, cap)), dim3(256), 0, st,
                           row_index_dev, column_index_dev, labels_dev, n, lut_dev, lut_size, h, w, winner, status_dev);
        PXSOM_LAUNCH_CHECK("mask_winner_kernel");
    }
    hipLaunchKernelGGL(mask_resolve_kernel, dim3((unsigned)std::min<int64_t>((pixels + 255) / 256, cap)), dim3(256), 0,
                       st, labels_dev, lut_dev, lut_size, pixels, winner, mask_dev);
    PXSOM_LAUNCH_CHECK("mask_resolve_kernel");
    return PXSOM_OK;
}

// ------------------------------------------------------------------------------------------------
// label -> label lookup (SOM cluster -> meta cluster): out[i] = lut[labels[i]], `fill` outside the table
// ------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void relabel_kernel(const int32_t *__restrict__ labels, int64_t n,
                                                      const int32_t *__restrict__ lut, int lut_size, int32_t fill,
                                                      int32_t *__restrict__ out)
{
    extern __shared__ int32_t lut_s[];
    for (int i = threadIdx.x; i < lut_size; i += 256) lut_s[i] = lut[i];
    __syncthreads();
    // 4 labels per thread and trip: 16-byte loads and stores when the vectors are aligned
    const int64_t n4 = ((reinterpret_cast<uintptr_t>(labels) | reinterpret_cast<uintptr_t>(out)) & 15) == 0 ? n / 4 : 0;
    typedef int v4 __attribute__((ext_vector_type(4)));
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < n4; v += (int64_t)gridDim.x * 256) {
        const v4 l = reinterpret_cast<const v4 *>(labels)[v];
        v4 o;
#pragma unroll
        for (int u = 0; u < 4; u++) o[u] = (l[u] >= 0 && l[u] < lut_size) ? lut_s[l[u]] : fill;
        reinterpret_cast<v4 *>(out)[v] = o;
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int l = labels[i];
        out[i] = (l >= 0 && l < lut_size) ? lut_s[l] : fill;
    }
}
}  // namespace

PXSOM_EXPORT int pxsom_relabel(const int32_t *labels_dev, int64_t n, const int32_t *lut_dev, int lut_size, int32_t fill,
                               int32_t *out_dev, void *stream)
{
    if (n < 0 || lut_size < 1 || lut_size > 16384 || !lut_dev || (n > 0 && (!labels_dev || !out_dev)))
        return pxsom::fail(PXSOM_ERR_INVALID_ARG, "pxsom_relabel: bad arguments (lookup tables of 1..16384 entries)");
    if (n == 0) return PXSOM_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t grid = std::min<int64_t>((n / 4 + 255) / 256 + 1, (int64_t)pxsom::device_cu_count() * 8);
    hipLaunchKernelGGL(relabel_kernel, dim3((unsigned)grid), dim3(256), (size_t)lut_size * sizeof(int32_t), st, labels_dev, n,
                       lut_dev, lut_size, fill, out_dev);
    PXSOM_LAUNCH_CHECK("relabel_kernel");
    return PXSOM_OK;
}
