"""ctypes binding of the CPU oracle (oracle/pxsom_oracle.c).  TEST INFRASTRUCTURE ONLY.

Imported by tests/, by ``__graft_entry__.smoke()`` and by ``bench.py``'s ``cpu_baseline`` leg --
never by anything under ``ark_analysis_amd/``.  Nothing here reads /root/reference.
"""
import ctypes
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE_DIR = os.path.join(_ROOT, "oracle")
_SO = os.path.join(_ORACLE_DIR, "libpxsom_oracle.so")
_lib = None

c_dp = ctypes.POINTER(ctypes.c_double)
c_i64p = ctypes.POINTER(ctypes.c_int64)
c_i32p = ctypes.POINTER(ctypes.c_int32)


def build(force: bool = False) -> str:
    src = os.path.join(_ORACLE_DIR, "pxsom_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _ORACLE_DIR, "-s", "-B"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        L.orc_som_online.restype = ctypes.c_int64
        L.orc_som_online.argtypes = [c_dp, ctypes.c_int64, ctypes.c_int, c_dp, ctypes.c_int, c_dp,
                                     ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                     ctypes.c_double, ctypes.c_int, c_i64p]
        L.orc_map_data_to_nodes.restype = ctypes.c_int
        L.orc_map_data_to_nodes.argtypes = [c_dp, ctypes.c_int, c_dp, ctypes.c_int64, ctypes.c_int,
                                            c_i32p, c_dp, ctypes.c_int]
        L.orc_nhbrdist_chebyshev.restype = None
        L.orc_nhbrdist_chebyshev.argtypes = [ctypes.c_int, ctypes.c_int, c_dp]
        L.orc_cluster_sums.restype = None
        L.orc_cluster_sums.argtypes = [c_dp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                       ctypes.c_int, c_i32p, ctypes.c_int, c_dp, c_i64p]
        L.orc_pair_histogram.restype = None
        L.orc_pair_histogram.argtypes = [c_i32p, c_i32p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, c_i64p]
        L.orc_batch_update.restype = None
        L.orc_batch_update.argtypes = [c_dp, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_dp, c_i64p,
                                       ctypes.c_double, ctypes.c_double]
        L.orc_batch_gain.restype = ctypes.c_double
        L.orc_batch_gain.argtypes = [ctypes.c_double, ctypes.c_double]
        L.orc_batch_gain_saturation.restype = ctypes.c_double
        L.orc_batch_gain_saturation.argtypes = [ctypes.c_double]
        L.orc_som_batch.restype = ctypes.c_int
        L.orc_som_batch.argtypes = [c_dp, ctypes.c_int64, ctypes.c_int, c_dp, ctypes.c_int, ctypes.c_int,
                                    ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                    ctypes.c_double, ctypes.c_int, ctypes.c_int]
        L.orc_som_batch_sched_q.restype = ctypes.c_int
        L.orc_som_batch_sched_q.argtypes = [c_dp, ctypes.c_int64, ctypes.c_int, c_dp, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                            ctypes.c_double, ctypes.c_int, ctypes.c_int, c_i32p, ctypes.c_int, ctypes.c_double]
        L.orc_gaussian_blur_hwc.restype = ctypes.c_int
        L.orc_gaussian_blur_hwc.argtypes = [c_dp, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_dp,
                                            ctypes.c_int]
        L.orc_gaussian_blur_hwc_ex.restype = ctypes.c_int
        L.orc_gaussian_blur_hwc_ex.argtypes = [c_dp, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_dp, ctypes.c_int,
                                               ctypes.c_int]
        L.orc_rowsum_filter_normalize.restype = ctypes.c_int64
        L.orc_rowsum_filter_normalize.argtypes = [c_dp, ctypes.c_int64, ctypes.c_int,
                                                  ctypes.c_double, ctypes.c_int, c_dp, c_i64p]
        L.orc_quantile_nonzero.restype = ctypes.c_double
        L.orc_quantile_nonzero.argtypes = [c_dp, ctypes.c_int64, ctypes.c_int64, ctypes.c_double,
                                           ctypes.c_int]
        L.orc_normalize_columns.restype = None
        L.orc_normalize_columns.argtypes = [c_dp, ctypes.c_int64, ctypes.c_int, c_dp, c_dp]
        L.orc_cluster_means.restype = None
        L.orc_cluster_means.argtypes = [c_dp, c_i64p, ctypes.c_int, ctypes.c_int, c_dp]
        L.orc_glibc_rand_fill.restype = None
        L.orc_glibc_rand_fill.argtypes = [ctypes.c_uint32, ctypes.c_int64, c_i32p]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(c_dp)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def nhbrdist(xdim, ydim):
    K = xdim * ydim
    out = np.empty((K, K), dtype=np.float64)
    lib().orc_nhbrdist_chebyshev(xdim, ydim, _dp(out))
    return out


# named switches for the RECALLED details of pyFlowSOM (oracle/pxsom_oracle.c ORC_V_*): 0 = the build's reading
V_COMPARE_SQUARED, V_NO_THRESHOLD_PIN, V_NO_EARLY_STOP, V_LAST_MINIMUM, V_INT_ABS = 1, 2, 4, 8, 16


def som_online(data, codes, xdim, ydim, rlen, alpha_range, radius_range, order, variant=0, node_order="xy"):
    """Returns trained codes [K, C] (copy).  ``order``: int64 [n*rlen].  ``variant`` / ``node_order``: the
    recalled-detail switches (tests of real pyFlowSOM vectors sweep them)."""
    data = _f64(data)
    codes = _f64(codes).copy()
    n, px = data.shape
    K = xdim * ydim
    assert codes.shape == (K, px)
    order = np.ascontiguousarray(order, dtype=np.int64)
    assert order.size == n * rlen and (n == 0 or (order.min() >= 0 and order.max() < n))
    nh = nhbrdist(xdim, ydim) if node_order == "xy" else _nhbrdist_yx(xdim, ydim)
    fn = lib().orc_som_online_ex
    fn.restype = ctypes.c_int64
    fn.argtypes = [c_dp, ctypes.c_int64, ctypes.c_int, c_dp, ctypes.c_int, c_dp, ctypes.c_double, ctypes.c_double,
                   ctypes.c_double, ctypes.c_double, ctypes.c_int, c_i64p, ctypes.c_int]
    steps = fn(_dp(data), n, px, _dp(codes), K, _dp(nh), float(alpha_range[0]), float(alpha_range[1]),
               float(radius_range[0]), float(radius_range[1]), int(rlen), order.ctypes.data_as(c_i64p), int(variant))
    assert steps >= 0
    return codes


def _nhbrdist_yx(xdim, ydim):
    """Chebyshev grid distances for the OTHER node numbering, k = y*xdim + x (switch of a recalled detail)."""
    gy, gx = np.divmod(np.arange(xdim * ydim), xdim)
    return np.ascontiguousarray(np.maximum(np.abs(gx[:, None] - gx[None, :]),
                                           np.abs(gy[:, None] - gy[None, :])).astype(np.float64))


def map_data_to_nodes_variant(codes, data, variant):
    codes, data = _f64(codes), _f64(data)
    n, px = data.shape
    labels = np.empty(n, dtype=np.int32)
    dists = np.empty(n, dtype=np.float64)
    fn = lib().orc_map_data_to_nodes_ex
    fn.restype = ctypes.c_int
    fn.argtypes = [c_dp, ctypes.c_int, c_dp, ctypes.c_int64, ctypes.c_int, c_i32p, c_dp, ctypes.c_int]
    assert fn(_dp(codes), codes.shape[0], _dp(data), n, px, labels.ctypes.data_as(c_i32p), _dp(dists), int(variant)) == 0
    return labels, dists


def map_data_to_nodes(codes, data, column_major_copy=False):
    """Returns (labels int32 1-based [n], dists f64 [n])."""
    codes, data = _f64(codes), _f64(data)
    if data.ndim == 1:
        data = data.reshape(0, codes.shape[1]) if data.size == 0 else data.reshape(1, -1)
    n, px = data.shape
    K = codes.shape[0]
    assert codes.shape[1] == px
    labels = np.empty(n, dtype=np.int32)
    dists = np.empty(n, dtype=np.float64)
    rc = lib().orc_map_data_to_nodes(_dp(codes), K, _dp(data), n, px,
                                     labels.ctypes.data_as(c_i32p), _dp(dists),
                                     1 if column_major_copy else 0)
    assert rc == 0
    return labels, dists


def cluster_sums(data, labels, K, row0=0, stride=1, count=None):
    data = _f64(data)
    labels = np.ascontiguousarray(labels, dtype=np.int32)
    n, px = data.shape
    if count is None:
        count = labels.size
    sums = np.zeros((K, px), dtype=np.float64)
    counts = np.zeros(K, dtype=np.int64)
    lib().orc_cluster_sums(_dp(data), row0, stride, count, px, labels.ctypes.data_as(c_i32p), K,
                           _dp(sums), counts.ctypes.data_as(c_i64p))
    return sums, counts


def pair_histogram(a, b, na, nb):
    a = np.ascontiguousarray(a, dtype=np.int32)
    b = np.ascontiguousarray(b, dtype=np.int32)
    hist = np.zeros((int(na), int(nb)), dtype=np.int64)
    lib().orc_pair_histogram(a.ctypes.data_as(c_i32p), b.ctypes.data_as(c_i32p), a.size, int(na), int(nb),
                             hist.ctypes.data_as(c_i64p))
    return hist


def cluster_means(sums, counts):
    sums = _f64(sums)
    counts = np.ascontiguousarray(counts, dtype=np.int64)
    K, C = sums.shape
    means = np.empty((K, C), dtype=np.float64)
    lib().orc_cluster_means(_dp(sums), counts.ctypes.data_as(c_i64p), K, C, _dp(means))
    return means


def batch_update(codes, xdim, ydim, sums, counts, thr, alpha):
    codes = _f64(codes).copy()
    K, px = codes.shape
    assert K == xdim * ydim
    sums = _f64(sums)
    counts = np.ascontiguousarray(counts, dtype=np.int64)
    lib().orc_batch_update(_dp(codes), int(xdim), int(ydim), px, _dp(sums), counts.ctypes.data_as(c_i64p),
                           float(thr), float(alpha))
    return codes


def batch_gain(den, q):
    """1 - q^den of the batch rule (orc_batch_gain: binary exponentiation in plain binary64 products)."""
    return float(lib().orc_batch_gain(float(den), float(q)))


def batch_gain_saturation(q):
    """Smallest power of two D from which on 1 - q^den is exactly 1 by construction (orc_batch_gain_saturation)."""
    return float(lib().orc_batch_gain_saturation(float(q)))


def som_batch(data, codes, xdim, ydim, rlen, alpha_range, radius_range, M):
    data = _f64(data)
    codes = _f64(codes).copy()
    n, px = data.shape
    assert codes.shape[0] == xdim * ydim
    rc = lib().orc_som_batch(_dp(data), n, px, _dp(codes), int(xdim), int(ydim), float(alpha_range[0]),
                             float(alpha_range[1]), float(radius_range[0]),
                             float(radius_range[1]), int(rlen), int(M))
    assert rc == 0
    return codes


def quantize(x, quantum):
    """Rows as they join the statistics of a reproducible run: rounded to multiples of ``quantum`` (half to even)."""
    x = _f64(x)
    return np.rint(x / quantum) * quantum if quantum else x


def som_batch_sched(data, codes, xdim, ydim, rlen, alpha_range, radius_range, phases, edges, quantum=0.0):
    """The scheduled batch rule (orc_som_batch_sched_q): step g of a pass takes the rows i with i % phases in
    [edges[g], edges[g+1]); ``quantum`` > 0: the rows join the sums rounded to its multiples."""
    data = _f64(data)
    codes = _f64(codes).copy()
    n, px = data.shape
    assert codes.shape[0] == xdim * ydim
    e = np.ascontiguousarray(np.asarray(edges, dtype=np.int32))
    rc = lib().orc_som_batch_sched_q(_dp(data), n, px, _dp(codes), int(xdim), int(ydim), float(alpha_range[0]),
                                     float(alpha_range[1]), float(radius_range[0]), float(radius_range[1]), int(rlen),
                                     int(phases), e.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), int(e.size - 1),
                                     float(quantum))
    assert rc == 0, "orc_som_batch_sched rejected the schedule"
    return codes


def gaussian_weights(sigma, truncate=4.0):
    """scipy.ndimage._filters._gaussian_kernel1d(sigma, 0, radius) restated with numpy ops
    (numpy's exp / pairwise sum are part of the reference numerics)."""
    radius = int(truncate * float(sigma) + 0.5)
    sigma2 = sigma * sigma
    x = np.arange(-radius, radius + 1)
    phi_x = np.exp(-0.5 / sigma2 * x ** 2)
    phi_x = phi_x / phi_x.sum()
    return phi_x, radius


def gaussian_blur_hwc(img, sigma, truncate=4.0, f32=False):
    img = _f64(img).copy()
    H, W, C = img.shape
    w, r = gaussian_weights(sigma, truncate)
    w = _f64(w[::-1])  # scipy hands correlate1d the reversed kernel (symmetric: same values)
    rc = lib().orc_gaussian_blur_hwc_ex(_dp(img), H, W, C, _dp(w), r, int(bool(f32)))
    assert rc == 0
    return img


def rowsum_filter_normalize(x, thresh, sum_mode=0):
    x = _f64(x)
    N, C = x.shape
    out = np.empty((N, C), dtype=np.float64)
    kept = np.empty(N, dtype=np.int64)
    m = lib().orc_rowsum_filter_normalize(_dp(x), N, C, float(thresh), int(sum_mode), _dp(out),
                                          kept.ctypes.data_as(c_i64p))
    return out[:m].copy(), kept[:m].copy()


def quantile_nonzero(col, q, keep_mode=0):
    col = _f64(col)
    return lib().orc_quantile_nonzero(_dp(col), col.size, 1, float(q), int(keep_mode))


def normalize_columns(x, norm):
    x, norm = _f64(x), _f64(norm)
    out = np.empty_like(x)
    lib().orc_normalize_columns(_dp(x), x.shape[0], x.shape[1], _dp(norm), _dp(out))
    return out


def glibc_rand(seed, count):
    out = np.empty(count, dtype=np.int32)
    lib().orc_glibc_rand_fill(int(seed) & 0xFFFFFFFF, count, out.ctypes.data_as(c_i32p))
    return out
