"""Parity of the HIP kernels (through the C ABI) with the CPU oracle.  `-m gpu` only.

Bars: BMU labels bit-exact; exact-online codebook bit-exact (asserted as equality of the
binary64 arrays); batch-rule codebook / per-cluster tables within 1e-12 relative (far inside
the 1e-5 of BASELINE.json's north_star; the slack is libm pow / atomic summation order).
"""
import numpy as np
import pytest
import torch

from ark_analysis_amd import som_device as sd
from ark_analysis_amd import synth
from ark_analysis_amd.flowsom import default_radius_range

pytestmark = pytest.mark.gpu


def _codebook(x, k, seed=3):
    rs = np.random.RandomState(seed)
    idx = rs.choice(x.shape[0], size=k, replace=x.shape[0] < k)
    return np.ascontiguousarray(x[idx].astype(np.float64))


def _gpu_assign(gpu, x, w, want_dists=False, screen_all_lists=False):
    xd = torch.from_numpy(x).to(gpu)
    wd = torch.from_numpy(w).to(gpu)
    labels, dists = sd.assign(xd, wd, want_dists=want_dists, screen_all_lists=screen_all_lists)
    torch.cuda.synchronize()
    return labels.cpu().numpy(), (dists.cpu().numpy() if dists is not None else None)


@pytest.mark.parametrize("n,c,k,dtype", [
    (100_003, 22, 100, np.float32),   # BASELINE config 2 shape (register-resident codebook path)
    (50_001, 22, 100, np.float64),    # reference's own dtype
    (20_000, 8, 100, np.float32),     # BASELINE config 1 shape
    (30_011, 16, 100, np.float32),    # fast path, 4 channels per lane
    (30_010, 30, 100, np.float64),    # fast path, 8 channels per lane
    (10_000, 12, 97, np.float32),     # fast path, K = 97
    (63, 22, 100, np.float32),        # below one 64-row group: generic path
    (4_097, 7, 100, np.float32),      # odd channel count: scalar-load path
    (3_000, 40, 400, np.float32),     # config 5 shape: two channel chunks, 25 node blocks
    (2_000, 100, 100, np.float32),    # config 4 (cell SOM) shape: four channel chunks
    (1_500, 15, 200, np.float64),     # reference test shape (20x10 grid)
    (777, 22, 25, np.float32),
    (64, 3, 4, np.float32),
    (1, 22, 100, np.float32),
    (5, 22, 1, np.float32),
    (30_000, 40, 400, np.float16),    # config 5: fp16 pixel matrix, 20x20 SOM
    (20_001, 22, 100, np.float16),    # fp16 rows on the register-resident path
    (2_000, 7, 30, np.float16),       # fp16, odd channel count
    (20_000, 48, 400, np.float16),    # packed-K fragments (binary16 rows, c % 8 == 0, K > 128): 3 MFMAs per block instead of 4
    (9_000, 72, 200, np.float16),     # packed: 5 instead of 6
    (5_000, 104, 144, np.float16),    # packed: 7 instead of 8
    (6_000, 64, 400, np.float16),     # c % 8 == 0 but nothing to gain: chunked layout
])
def test_assign_matches_oracle(gpu, oracle, n, c, k, dtype):
    x = synth.make_fov_numpy(max(n, 2 * k), c, seed=11, dtype=dtype)[:n]
    w = _codebook(synth.make_fov_numpy(4 * k + 50, c, seed=12, dtype=np.float64), k)
    w += 1e-3 * np.random.RandomState(1).standard_normal(w.shape)
    got, _ = _gpu_assign(gpu, x, w)
    want, _ = oracle.map_data_to_nodes(w, x.astype(np.float64))
    assert got.dtype == np.int32 and got.shape == (n,)
    np.testing.assert_array_equal(got, want)
    assert got.min() >= 1 and got.max() <= k


@pytest.mark.parametrize("seed", range(24))
def test_assign_random_shapes_against_oracle(gpu, oracle, seed):
    """Seeded sweep over shapes / dtypes / degeneracies nobody hand-picked: every label must equal the
    oracle's (first strict minimum), whatever path (fast / generic filter, exact) the shape takes."""
    rs = np.random.RandomState(1000 + seed)
    n = int(rs.choice([1, 2, 63, 64, 65, 257, 1000, 4097, 20_000]))
    c = int(rs.choice([1, 2, 3, 8, 15, 22, 32, 33, 40, 64, 100, 128]))
    k = int(rs.choice([1, 2, 16, 17, 97, 100, 128, 200, 400, 1024]))
    dtype = [np.float16, np.float32, np.float64][seed % 3]
    scale = float(rs.choice([1e-3, 1.0, 50.0])) if dtype != np.float16 else 1.0
    x = (rs.gamma(0.7, 0.4, size=(n, c)) * scale).astype(dtype)
    x[rs.uniform(size=x.shape) < 0.15] = 0
    w = x[rs.randint(0, n, size=k)].astype(np.float64)              # nodes are data rows: exact zero distances
    w += (rs.uniform(size=(k, 1)) < 0.5) * 1e-3 * scale * rs.standard_normal((k, c))
    if k > 3:
        w[k // 2] = w[0]                                                # duplicate node: the lower index must win
        w[k - 1] = np.nextafter(w[1], np.inf)                           # one-ulp neighbour
    if n > 10:
        x[3] = x[7]
        x[5] = 0
    got, _ = _gpu_assign(gpu, x, w)
    want, _ = oracle.map_data_to_nodes(w, x.astype(np.float64))
    np.testing.assert_array_equal(got, want)


def test_assign_empty(gpu):
    x = torch.empty((0, 22), dtype=torch.float32, device=gpu)
    w = torch.rand((100, 22), dtype=torch.float64, device=gpu)
    labels, _ = sd.assign(x, w)
    assert labels.numel() == 0


def test_assign_exact_ties_and_near_ties(gpu, oracle):
    """Duplicate codebook rows (exact ties -> first index wins), rows sitting exactly on nodes,
    and rows within a few ulp of the midpoint between two nodes."""
    c, k = 22, 100
    base = synth.make_fov_numpy(5000, c, seed=21, dtype=np.float32)
    w = _codebook(base.astype(np.float64), k, seed=5)
    w[37] = w[3]          # exact duplicates: label must be 4, never 38
    w[99] = w[98]
    rows = [base[:2000]]
    rows.append(w[[3, 37, 98, 99, 0, 50]].astype(np.float32))          # rows (nearly) on nodes
    mid = (0.5 * (w[10] + w[11]))[None, :].repeat(200, axis=0)
    mid += 1e-7 * np.random.RandomState(2).standard_normal(mid.shape)   # near-ties
    rows.append(mid.astype(np.float32))
    x = np.ascontiguousarray(np.concatenate(rows))
    for dtype in (np.float32, np.float64):
        xx = x.astype(dtype)
        got, _ = _gpu_assign(gpu, xx, w)
        want, _ = oracle.map_data_to_nodes(w, xx.astype(np.float64))
        np.testing.assert_array_equal(got, want)
    assert 38 not in got and 100 not in got


def test_assign_identical_codebook_rows(gpu, oracle):
    """Degenerate codebooks.  Exact duplicates of an earlier node are masked in the filter (they can
    never win under first-minimum tie-break), so an all-identical codebook is resolved without the
    exact path; nodes that differ by one ulp in one channel are NOT duplicates and every row whose
    scores tie within the error bound must take the exact path."""
    x = synth.make_fov_numpy(3000, 22, seed=4)
    w = np.tile(x[:1].astype(np.float64), (100, 1))
    got, _ = _gpu_assign(gpu, x, w)
    np.testing.assert_array_equal(got, np.ones(3000, dtype=np.int32))
    assert sd.last_exact_rows(sd.assign.last_workspace) < 300
    w2 = w.copy()
    w2[1:, 3] = np.nextafter(w2[1:, 3], 10.0) + np.arange(99) * 1e-15   # all distinct, all near-ties
    got, _ = _gpu_assign(gpu, x, w2)
    want, _ = oracle.map_data_to_nodes(w2, x.astype(np.float64))
    np.testing.assert_array_equal(got, want)
    assert sd.last_exact_rows(sd.assign.last_workspace) >= 3000


@pytest.mark.parametrize("n,c,k,dtype", [
    (40_000, 22, 100, np.float32),     # config 2's codebook
    (30_000, 40, 400, np.float16),     # config 5's
    (30_000, 100, 100, np.float32),    # config 4's
    (2_500, 100, 100, np.float32),     # ... a short list: sixteen lane groups share the nodes of four rows per wave
    (1_500, 40, 400, np.float16),      # config 5's, short list
    (6_000, 100, 100, np.float64),     # binary64 rows with the LDS-staged copy (rows re-read, not taken from registers)
    (20_000, 7, 97, np.float64),       # odd channel count, binary64 rows
    (12_000, 128, 225, np.float32),    # the widest rows
    (9_000, 3, 1024, np.float32),      # the largest codebook
])
@pytest.mark.parametrize("pattern", ["crowded", "ties", "wild"])
def test_long_exact_lists_are_screened_and_stay_bit_exact(gpu, oracle, n, c, k, dtype, pattern):
    """Thousands of listed rows (crowded codebooks, discrete data, non-finite and out-of-range rows) through the
    screened exact kernel (binary32 screening against the distance of the filter's proposal, binary64 only for
    the surviving nodes): labels equal the oracle's, first-minimum ties and label 0 for non-finite rows included.
    The kernel takes over from 2.25e6 / C listed rows; PXSOM_ASSIGN_SCREEN_ALL_LISTS (pxsom_assign_ex) sends these lists there too."""
    rs = np.random.RandomState(n + c + k)
    if pattern == "crowded":       # node pairs 1e-3 .. 1e-9 apart, a duplicated node, rows around them
        half = rs.rand((k + 1) // 2, c)
        w = np.concatenate([half, half * (1.0 + 10.0 ** rs.uniform(-9, -3, size=(len(half), 1)))])[:k]
        w[k - 1] = w[0]
        x = w[rs.randint(0, k, n)] * (1.0 + 1e-4 * rs.standard_normal((n, c)))
    elif pattern == "ties":        # values on a coarse grid: exact distance ties between nodes, zero distances
        x = rs.randint(0, 3, size=(n, c)).astype(np.float64) / 2.0
        w = rs.randint(0, 3, size=(k, c)).astype(np.float64) / 2.0
        w[: min(k, 50)] = x[: min(k, 50)]
    else:                          # one blob (every row near-tied) with rows no shortcut survives
        x = 0.5 + 0.01 * rs.standard_normal((n, c))
        w = 0.5 + 0.01 * rs.standard_normal((k, c))
        for value in (np.nan, np.inf, -np.inf, 1e30, -1e30, 1e-30, 6e4, 7e4, 1e19, 3e38):
            hit = rs.randint(0, n, 40)
            x[hit, rs.randint(0, c, 40)] = value
        x[7] = np.nan
        x[8] = 0.0
    with np.errstate(over="ignore"):
        x = np.ascontiguousarray(x.astype(dtype))
    w = np.ascontiguousarray(w.astype(np.float64))
    got, _ = _gpu_assign(gpu, x, w, screen_all_lists=True)
    if pattern != "ties":          # (coarse-grid ties list thousands of rows for most shapes, not for all)
        # (the filters centre rows and codebook: an offset blob is no longer "every row near-tied" for them -- the rows no
        # shortcut survives remain)
        assert sd.last_exact_rows(sd.assign.last_workspace) >= (min(2048, n // 2) if pattern == "crowded" else 256)
    want, _ = oracle.map_data_to_nodes(w, x.astype(np.float64))
    np.testing.assert_array_equal(got, want)
    got, _ = _gpu_assign(gpu, x, w)                    # and the default split between the two exact kernels
    np.testing.assert_array_equal(got, want)


def test_assign_nonfinite_rows(gpu, oracle):
    x = synth.make_fov_numpy(1000, 22, seed=6)
    x[5, 3] = np.nan
    x[17, :] = np.nan
    x[40, 0] = np.inf
    x[41, 2] = 1e30          # finite but far outside fp16 range: exact path, ordinary label
    w = _codebook(x[100:].astype(np.float64), 100)
    got, _ = _gpu_assign(gpu, x, w)
    want, _ = oracle.map_data_to_nodes(w, x.astype(np.float64))
    np.testing.assert_array_equal(got, want)
    assert got[5] == 0 and got[17] == 0 and got[40] == 0 and got[41] >= 1


def test_assign_large_and_tiny_magnitudes(gpu, oracle):
    """Un-normalised scales (cell tables: counts / areas) must not break the fp16 filter."""
    rs = np.random.RandomState(8)
    for scale in (1e-6, 1e4):
        x = (synth.make_fov_numpy(5000, 16, seed=9, dtype=np.float64) * scale)
        w = _codebook(x, 100) * (1 + 1e-3 * rs.standard_normal((100, 16)))
        got, _ = _gpu_assign(gpu, x, w)
        want, _ = oracle.map_data_to_nodes(w, x)
        np.testing.assert_array_equal(got, want)


def test_assign_strided_rows_and_dists(gpu, oracle):
    """ldx > c (mini-batch views x[t::M]) and the optional distance output."""
    x = synth.make_fov_numpy(40_000, 22, seed=13)
    w = _codebook(x.astype(np.float64), 100)
    xd = torch.from_numpy(x).to(gpu)
    wd = torch.from_numpy(w).to(gpu)
    view = xd[3::7]
    assert not view.is_contiguous()
    labels, dists = sd.assign(view, wd, want_dists=True)
    want_l, want_d = oracle.map_data_to_nodes(w, x[3::7].astype(np.float64))
    np.testing.assert_array_equal(labels.cpu().numpy(), want_l)
    np.testing.assert_array_equal(dists.cpu().numpy(), want_d)


def test_cluster_sums_matches_oracle(gpu, oracle):
    for (n, c, k, dtype) in [(100_000, 22, 100, np.float32), (30_000, 40, 400, np.float64),
                             (999, 5, 7, np.float32), (50_001, 22, 100, np.float32),
                             (20_000, 40, 400, np.float16)]:
        x = synth.make_fov_numpy(n, c, seed=14, dtype=dtype)
        labels = np.random.RandomState(1).randint(0, k + 1, size=n).astype(np.int32)  # 0 = skipped
        s, cnt = sd.cluster_sums(torch.from_numpy(x).to(gpu), torch.from_numpy(labels).to(gpu), k)
        ws, wc = oracle.cluster_sums(x.astype(np.float64), labels, k)
        np.testing.assert_array_equal(cnt.cpu().numpy(), wc)
        np.testing.assert_allclose(s.cpu().numpy(), ws, rtol=1e-12, atol=1e-12)


def test_cluster_mask_kernel_against_numpy(gpu):
    """pxsom_cluster_mask on a 1024 x 1024 image: unique pixels, repeated pixels (last row wins), unmapped
    labels and stray coordinates reported through the status word."""
    rs = np.random.RandomState(21)
    h, w, k = 1024, 1024, 300
    pos = rs.permutation(h * w)[:900_000]
    pos = np.concatenate([pos, pos[:50_000]])                  # 50 000 pixels listed twice
    labels = rs.randint(0, k, size=pos.size)
    lut = rs.randint(-40_000, 40_000, size=k).astype(np.int32)
    narrowed = lut.astype(np.int16).astype(np.int32)

    def dev(a, dt):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(gpu)
    mask, status = sd.cluster_mask(dev(pos // w, np.int64), dev(pos % w, np.int64), dev(labels, np.int64),
                                   dev(narrowed, np.int32), h, w)
    want = np.zeros(h * w, dtype=np.int16)
    want[pos] = lut[labels].astype(np.int16)                   # numpy assigns in order: the last row stays
    assert status == 0
    np.testing.assert_array_equal(mask.cpu().numpy().ravel(), want)
    holes = narrowed.copy()
    holes[7] = sd.LUT_UNMAPPED
    assert sd.cluster_mask(dev(pos // w, np.int64), dev(pos % w, np.int64), dev(labels, np.int64),
                           dev(holes, np.int32), h, w)[1] == sd.MASK_BAD_LABEL
    assert sd.cluster_mask(dev(pos // w, np.int64), dev(pos % w, np.int64), dev(labels + 1, np.int64),
                           dev(narrowed, np.int32), h, w)[1] == sd.MASK_BAD_LABEL          # label k: past the LUT
    rows = pos // w
    rows[5] = h
    assert sd.cluster_mask(dev(rows, np.int64), dev(pos % w, np.int64), dev(labels, np.int64),
                           dev(narrowed, np.int32), h, w)[1] == sd.MASK_BAD_PIXEL
    empty = torch.empty(0, dtype=torch.int64, device=gpu)
    mask, status = sd.cluster_mask(empty, empty, empty, dev(narrowed, np.int32), 5, 7)
    assert status == 0 and not mask.any() and mask.shape == (5, 7)


@pytest.mark.parametrize("n,c,k,dtype,mode", [
    (300_001, 22, 100, np.float32, "uniform"),   # 2 rows per instruction, 8 tables per CU
    (300_001, 22, 100, np.float32, "runs"),      # equal neighbouring labels: the row-by-row path
    (200_000, 22, 100, np.float32, "one"),       # every row in one cluster
    (100_003, 13, 50, np.float32, "uniform"),    # 4 rows per instruction
    (100_003, 21, 100, np.float32, "runs"),      # 3 rows per instruction (label registers of 60 rows)
    (100_003, 33, 64, np.float32, "uniform"),    # 1 row per instruction
    (100_003, 64, 100, np.float16, "uniform"),   # 2 tables per CU
    (100_003, 22, 100, np.float64, "uniform"),
    (32_768, 16, 8, np.float32, "skew"),          # channel pairs per lane: 8 rows per instruction
    (100_003, 40, 100, np.float32, "runs"),       # ... 3 rows per instruction
    (100_003, 32, 100, np.float32, "uniform"),    # ... 4
    (150_001, 22, 100, np.float16, "runs"),       # ... 5, fp16 pairs
    (100_003, 14, 30, np.float32, "skew"),        # ... 8 (capped), idle lanes
])
def test_cluster_sums_wave_private_tables(gpu, oracle, n, c, k, dtype, mode):
    """Shapes served by the wave-private-table kernels (13 <= c <= 64, n >= 32768; fp32 / fp16 rows with an even
    channel count take the channel-pair form of pxsom_sums.hip, the rest one channel per lane).  The values are multiples
    of 2^-8 below 4, so every partial sum is exact in binary64 and the order of the additions cannot show:
    bit-equal to the oracle."""
    rs = np.random.RandomState(n % 1000 + c)
    x = (rs.randint(0, 1024, size=(n, c)) / 256.0).astype(dtype)
    if mode == "uniform":
        labels = rs.randint(0, k + 2, size=n)          # 0 and k + 1 are skipped
    elif mode == "runs":
        labels = np.arange(n) // 3 % k + 1
    elif mode == "one":
        labels = np.full(n, 7)
    else:
        labels = np.where(rs.rand(n) < 0.7, 1, rs.randint(1, k + 1, size=n))
    labels = labels.astype(np.int32)
    xd = torch.from_numpy(x).to(gpu)
    s, cnt = sd.cluster_sums(xd, torch.from_numpy(labels).to(gpu), k)
    inside = np.where((labels >= 1) & (labels <= k), labels, 0)
    ws, wc = oracle.cluster_sums(x.astype(np.float64), inside, k)
    np.testing.assert_array_equal(cnt.cpu().numpy(), wc)
    np.testing.assert_array_equal(s.cpu().numpy(), ws)
    # a column window of a wider matrix (ldx > c) accumulating on top of the first result
    wide = torch.from_numpy(np.concatenate([x, x], axis=1)).to(gpu)
    s2, cnt2 = sd.cluster_sums(wide[:, c:], torch.from_numpy(labels).to(gpu), k, sums=s, counts=cnt)
    np.testing.assert_array_equal(cnt2.cpu().numpy(), 2 * wc)
    np.testing.assert_array_equal(s2.cpu().numpy(), 2 * ws)


@pytest.mark.parametrize("n,c,xdim,ydim,rlen,dtype", [
    (20_000, 22, 10, 10, 1, np.float32),
    (5_000, 22, 10, 10, 2, np.float64),
    (3_000, 8, 10, 10, 1, np.float32),
    (2_000, 15, 20, 10, 1, np.float64),   # reference test grid (cluster_helpers_test: xdim=20, ydim=10)
    (1_000, 40, 20, 20, 1, np.float32),   # config 5 grid
    (300, 4, 3, 2, 3, np.float32),
    (2_000, 20, 8, 8, 1, np.float32),     # 4 lanes per node, 6 channels per lane
    (1_500, 33, 6, 5, 2, np.float64),     # 4 lanes per node, 10 channels per lane
    (2_000, 5, 10, 10, 1, np.float64),    # 2 lanes per node, 4 channels per lane
    (2_000, 14, 11, 11, 1, np.float32),   # 2 lanes per node, 8 channels per lane
    (1_500, 37, 10, 12, 1, np.float32),   # 2 lanes per node, 20 channels per lane
    (1_000, 45, 10, 10, 1, np.float32),   # wide rows: 4 lanes per node in 512 threads, 16 channels per lane
    (1_000, 70, 9, 11, 2, np.float64),    # ... 20 channels per lane
    (1_500, 100, 10, 10, 1, np.float32),  # ... 26 channels per lane: the cell SOM of config 4
    (800, 104, 8, 16, 1, np.float64),     # ... all 104 slots used, 128 nodes
    (1_000, 70, 12, 12, 1, np.float64),   # thread <-> node form (more than 128 nodes), 104 registers per node
    (600, 100, 16, 16, 1, np.float32),    # thread <-> node form, 256 nodes
    (500, 110, 10, 10, 1, np.float32),    # thread <-> node form, codebook in LDS (c > 104)
    (2_000, 22, 10, 10, 1, np.float16),   # fp16 rows
])
def test_train_online_bit_exact(gpu, oracle, n, c, xdim, ydim, rlen, dtype):
    k = xdim * ydim
    x = synth.make_fov_numpy(max(n, k), c, seed=15, dtype=dtype)[:n]
    rs = np.random.RandomState(16)
    w0 = np.ascontiguousarray(x[rs.choice(n, k, replace=n < k)].astype(np.float64))
    order = rs.randint(0, n, size=n * rlen).astype(np.int64)
    ar, rr = (0.05, 0.01), default_radius_range(xdim, ydim)
    want = oracle.som_online(x.astype(np.float64), w0, xdim, ydim, rlen, ar, rr, order)
    wd = torch.from_numpy(w0.copy()).to(gpu)
    sd.train_online(torch.from_numpy(x).to(gpu), wd, xdim, ydim, rlen, ar, rr,
                    torch.from_numpy(order).to(gpu))
    got = wd.cpu().numpy()
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("n,c,xdim,ydim,rlen,dtype", [
    (1_500, 22, 10, 10, 2, np.float32),   # 2 lanes per node
    (900, 8, 6, 6, 3, np.float64),        # 4 lanes per node
    (700, 70, 10, 10, 2, np.float32),     # wide rows, 4 lanes per node in 512 threads
    (600, 30, 14, 14, 2, np.float32),     # thread <-> node form
    (500, 110, 10, 10, 2, np.float32),    # codebook in LDS
])
def test_train_online_early_stop_fires(gpu, oracle, n, c, xdim, ydim, rlen, dtype):
    """FlowSOM's "if (change < 1) stop at the start of a pass" (one more step runs, then the loop ends), with BOTH readings
    of its accumulator (oracle ORC_V_INT_ABS / pxsom_train_online_ex PXSOM_ONLINE_INT_ABS): the integer abs() adds nothing
    while every |x - w| < 1, so the run stops at the start of its second pass; fabs stops only when a whole pass moved the
    codebook by less than 1 in total (tiny data).  Codebooks bit-equal to the oracle's in all four combinations."""
    k = xdim * ydim
    x = np.minimum(synth.make_fov_numpy(max(n, k), c, seed=25, dtype=np.float32)[:n], 0.9).astype(dtype)
    rs = np.random.RandomState(26)
    w0 = np.ascontiguousarray(x[rs.choice(n, k, replace=n < k)].astype(np.float64))
    order = rs.randint(0, n, size=n * rlen).astype(np.int64)
    ar, rr = (0.05, 0.01), default_radius_range(xdim, ydim)
    xd, od = torch.from_numpy(x).to(gpu), torch.from_numpy(order).to(gpu)
    outs = {}
    for int_abs in (False, True):
        want = oracle.som_online(x.astype(np.float64), w0, xdim, ydim, rlen, ar, rr, order, variant=oracle.V_INT_ABS if int_abs else 0)
        wd = torch.from_numpy(w0.copy()).to(gpu)
        sd.train_online(xd, wd, xdim, ydim, rlen, ar, rr, od, int_abs=int_abs)
        np.testing.assert_array_equal(wd.cpu().numpy(), want)
        outs[int_abs] = want
    assert not np.array_equal(outs[False], outs[True])        # the stop fired under the integer reading only
    # a data set so small in magnitude that fabs stops too: the two readings agree again
    tiny = (x.astype(np.float64) * 1e-7).astype(dtype) if dtype != np.float32 else (x * np.float32(1e-7))
    w0t = np.ascontiguousarray(tiny[rs.choice(n, k, replace=n < k)].astype(np.float64))
    want = oracle.som_online(tiny.astype(np.float64), w0t, xdim, ydim, rlen, ar, rr, order)
    stopped = oracle.som_online(tiny.astype(np.float64), w0t, xdim, ydim, rlen, ar, rr, order, variant=oracle.V_NO_EARLY_STOP)
    assert not np.array_equal(want, stopped)                  # (the early stop did fire in the default reading)
    wd = torch.from_numpy(w0t.copy()).to(gpu)
    sd.train_online(torch.from_numpy(tiny).to(gpu), wd, xdim, ydim, rlen, ar, rr, od)
    np.testing.assert_array_equal(wd.cpu().numpy(), want)


@pytest.mark.parametrize("c,xdim,ydim", [(22, 10, 10), (6, 7, 9), (12, 5, 5), (30, 16, 16), (100, 10, 10), (50, 8, 8)])
def test_train_online_ties_bit_exact(gpu, oracle, c, xdim, ydim):
    """Coarsely quantised rows and duplicated initial nodes: equal and near-equal distances are the
    rule, so the first-strict-minimum path (sqrt comparison, lowest node wins) is what is tested."""
    k = xdim * ydim
    n = 4_000
    rs = np.random.RandomState(61)
    x = rs.randint(0, 3, size=(n, c)).astype(np.float32)
    x[::11] = 0.0
    w0 = np.ascontiguousarray(x[rs.randint(0, n // 50, size=k)].astype(np.float64))
    order = rs.randint(0, n, size=n).astype(np.int64)
    ar, rr = (0.05, 0.01), default_radius_range(xdim, ydim)
    want = oracle.som_online(x.astype(np.float64), w0, xdim, ydim, 1, ar, rr, order)
    wd = torch.from_numpy(w0.copy()).to(gpu)
    sd.train_online(torch.from_numpy(x).to(gpu), wd, xdim, ydim, 1, ar, rr, torch.from_numpy(order).to(gpu))
    np.testing.assert_array_equal(wd.cpu().numpy(), want)


@pytest.mark.parametrize("c,xdim,ydim", [(22, 10, 10), (20, 8, 8)])
def test_train_online_near_ties_take_the_exact_path(gpu, oracle, c, xdim, ydim):
    """Pairs of initial nodes a few ulp apart: their distances differ, but by far less than the key step the
    fast (tree-summed) minimum may decide, and in an order a reassociated sum could flip -- the winner has to
    come from the left-to-right sums.  Bit-equal codebook after two epochs."""
    k = xdim * ydim
    n = 3_000
    x = synth.make_fov_numpy(n, c, seed=19, dtype=np.float32)
    rs = np.random.RandomState(62)
    w0 = np.ascontiguousarray(x[rs.choice(n, size=k, replace=False)].astype(np.float64))
    for a in range(0, k - 1, 2):                       # node a+1 = node a nudged by 1..3 ulp per channel
        w0[a + 1] = w0[a]
        for _ in range(3):
            w0[a + 1] = np.where(rs.rand(c) < 0.5, np.nextafter(w0[a + 1], 2.0), w0[a + 1])
    order = rs.randint(0, n, size=2 * n).astype(np.int64)
    ar, rr = (0.05, 0.01), (0.4, 0.0)                  # threshold 0.5 throughout: only the winner moves
    want = oracle.som_online(x.astype(np.float64), w0, xdim, ydim, 2, ar, rr, order)
    wd = torch.from_numpy(w0.copy()).to(gpu)
    sd.train_online(torch.from_numpy(x).to(gpu), wd, xdim, ydim, 2, ar, rr, torch.from_numpy(order).to(gpu))
    np.testing.assert_array_equal(wd.cpu().numpy(), want)


@pytest.mark.parametrize("n,c,k,dtype,stride", [
    (16_384, 22, 100, np.float32, 1),    # fused route: filter + exact accumulate (config 2 mini-batch)
    (16_391, 22, 100, np.float32, 64),   # strided mini-batch view x[t::64], ragged last group
    (5_003, 22, 100, np.float64, 1),     # fp64 table
    (4_097, 16, 98, np.float32, 3),      # 4 channels per lane
    (3_001, 30, 100, np.float64, 1),     # 8 channels per lane
    (2_000, 6, 100, np.float32, 1),      # 2 channels per lane
    (2_000, 40, 400, np.float32, 1),     # unfused route (generic filter + cluster sums)
    (4_000, 40, 400, np.float16, 1),     # config 5 dtype, unfused route
    (6_001, 22, 100, np.float16, 2),     # fp16 rows on the fused route
    (50, 22, 100, np.float32, 1),        # unfused: fewer than 64 rows
])
def test_batch_accumulate_matches_oracle(gpu, oracle, n, c, k, dtype, stride):
    """One mini-batch step's accumulation half: labels bit-exact, counts exact, sums to 1e-12.
    Rows are duplicated / quantised / NaN-poisoned so that listed rows (exact kernel) take part."""
    rs = np.random.RandomState(77)
    big = synth.make_fov_numpy(n * stride, c, seed=31, dtype=dtype)
    big[::17] = np.round(big[::17] * 2) / 2          # coarse rows: near-ties
    big[5 * stride] = np.nan                         # one NaN row inside the view
    w = _codebook(big[::stride], k, seed=5)
    w[7] = w[3]                                      # duplicate node
    w[11] = (w[12] + w[13]) / 2
    xd_big = torch.from_numpy(big).to(gpu)
    xv = xd_big[::stride]
    wd = torch.from_numpy(w).to(gpu)
    labels = torch.empty(n, dtype=torch.int32, device=gpu)
    stats = torch.full((k * (c + 1),), 123.0, dtype=torch.float64, device=gpu)   # must be cleared
    ws = sd.AssignWorkspace(n, c, k, gpu)
    sd.batch_accumulate(xv, wd, labels, stats, ws)
    xh = np.ascontiguousarray(big[::stride]).astype(np.float64)
    want_l, _ = oracle.map_data_to_nodes(w, xh)
    np.testing.assert_array_equal(labels.cpu().numpy(), want_l)
    ws_, wc_ = oracle.cluster_sums(np.nan_to_num(xh), want_l, k)
    got = stats.cpu().numpy()
    np.testing.assert_array_equal(got[k * c:], wc_.astype(np.float64))
    np.testing.assert_allclose(got[:k * c].reshape(k, c), ws_, rtol=1e-12, atol=1e-12)
    # a second call on the same buffers gives the same answer (statistics are cleared, not added to)
    sd.batch_accumulate(xv, wd, labels, stats, ws)
    np.testing.assert_allclose(stats.cpu().numpy(), got, rtol=1e-12, atol=1e-12)


def test_pair_histogram_matches_oracle(gpu, oracle):
    rs = np.random.RandomState(9)
    for n, na, nb in [(1_000_003, 5_000, 100), (50_000, 7, 3), (10, 1, 1), (0, 4, 4)]:
        a = rs.randint(-2, na + 3, size=n).astype(np.int32)      # some pairs fall outside and are ignored
        b = rs.randint(-1, nb + 2, size=n).astype(np.int32)
        got = sd.pair_histogram(torch.from_numpy(a).to(gpu), torch.from_numpy(b).to(gpu), na, nb)
        np.testing.assert_array_equal(got.cpu().numpy(), oracle.pair_histogram(a, b, na, nb))


def test_batch_update_matches_oracle(gpu, oracle):
    rs = np.random.RandomState(17)
    for (xdim, ydim, c) in [(10, 10, 22), (20, 20, 40), (3, 2, 4)]:
        k = xdim * ydim
        w = rs.uniform(size=(k, c))
        sums = rs.uniform(size=(k, c)) * 50
        counts = rs.randint(0, 100, size=k).astype(np.int64)
        counts[::7] = 0
        for thr, alpha in [(6.0, 0.05), (1.2, 0.03), (0.5, 0.01)]:
            want = oracle.batch_update(w, xdim, ydim, sums, counts, thr, alpha)
            wd = torch.from_numpy(w.copy()).to(gpu)
            sd.batch_update(wd, xdim, ydim, torch.from_numpy(sums).to(gpu),
                            torch.from_numpy(counts.astype(np.float64)).to(gpu), thr, alpha)
            np.testing.assert_allclose(wd.cpu().numpy(), want, rtol=1e-12, atol=0)


@pytest.mark.parametrize("xdim,ydim,c,dtype", [(10, 10, 22, np.float32), (10, 10, 16, np.float64),
                                               (7, 9, 12, np.float32), (20, 20, 40, np.float32)])
def test_batch_update_prepare_matches_oracle_and_chains(gpu, oracle, xdim, ydim, c, dtype):
    """update_prepare == oracle batch update; statistics cleared; an accumulate on the prepared
    workspace equals an unprepared one (labels bit-exact, statistics to 1e-12)."""
    from ark_analysis_amd.distributed import BatchSOMTrainer
    k, n = xdim * ydim, 9_000
    rs = np.random.RandomState(5)
    x = synth.make_fov_numpy(n, c, seed=41, dtype=dtype)
    w = _codebook(x, k, seed=8)
    sums = rs.uniform(size=(k, c)) * 40
    counts = rs.randint(0, 90, size=k).astype(np.float64)
    counts[::9] = 0
    xd = torch.from_numpy(x).to(gpu)
    ws = sd.AssignWorkspace(n, c, k, gpu)
    for thr, alpha in [(6.0, 0.05), (1.4, 0.02), (0.5, 0.01)]:
        want_w = oracle.batch_update(w, xdim, ydim, sums, counts.astype(np.int64), thr, alpha)
        wd = torch.from_numpy(w.copy()).to(gpu)
        stats = torch.from_numpy(np.concatenate([sums.reshape(-1), counts])).to(gpu)
        # two alternating buffers: the launch clears the OTHER one and leaves the one it read alone
        other = torch.full_like(stats, 7.0)
        wd2 = torch.from_numpy(w.copy()).to(gpu)
        sd.batch_update_prepare(wd2, xdim, ydim, stats, thr, alpha, ws, stats_next=other)
        np.testing.assert_allclose(wd2.cpu().numpy(), want_w, rtol=1e-12, atol=0)
        assert float(other.abs().max()) == 0.0 and float(stats.abs().max()) > 0.0
        # single buffer: cleared after the update
        sd.batch_update_prepare(wd, xdim, ydim, stats, thr, alpha, ws)
        np.testing.assert_allclose(wd.cpu().numpy(), want_w, rtol=1e-12, atol=0)
        assert float(stats.abs().max()) == 0.0
        labels = torch.empty(n, dtype=torch.int32, device=gpu)
        sd.batch_accumulate(xd, wd, labels, stats, ws, prepared=True)
        labels2 = torch.empty(n, dtype=torch.int32, device=gpu)
        stats2 = torch.empty_like(stats)
        sd.batch_accumulate(xd, wd, labels2, stats2, sd.AssignWorkspace(n, c, k, gpu))
        assert torch.equal(labels, labels2)
        want_l, _ = oracle.map_data_to_nodes(wd.cpu().numpy(), x.astype(np.float64))
        np.testing.assert_array_equal(labels.cpu().numpy(), want_l)
        np.testing.assert_allclose(stats.cpu().numpy(), stats2.cpu().numpy(), rtol=1e-12, atol=1e-12)
    if dtype != np.float32:
        # binary64 rows: the per-cluster sums are atomics in arbitrary order (1e-16 relative noise), which
        # the degenerate early steps of a 4-step pass amplify (BMU flips between near-identical nodes);
        # fp32 rows sum exactly in binary64, so only they give a reproducible whole-pass comparison
        return
    # a trainer reused after its codebook was reset must not reuse stale prepared state
    tr = BatchSOMTrainer(xdim, ydim, c, gpu, batch_steps=4)
    w0 = torch.from_numpy(w.copy()).to(gpu)
    wa = tr.train(xd, w0.clone(), num_passes=1).clone()
    wb = w0.clone()
    tr.train(xd, wb, num_passes=1)
    np.testing.assert_allclose(wb.cpu().numpy(), wa.cpu().numpy(), rtol=1e-12, atol=0)
    want_b = oracle.som_batch(x.astype(np.float64), w, xdim, ydim, 1, (0.05, 0.01),
                              default_radius_range(xdim, ydim), 4)
    np.testing.assert_allclose(wa.cpu().numpy(), want_b, rtol=1e-9, atol=0)


@pytest.mark.parametrize("c,dtype,n,m,passes,grid", [
    (22, np.float32, 40_000, 8, 1, 10), (22, np.float32, 9_001, 4, 2, 10), (8, np.float32, 20_000, 16, 1, 10),
    (16, np.float16, 30_000, 8, 1, 10), (32, np.float32, 12_345, 5, 1, 10), (22, np.float64, 10_000, 4, 1, 10),
    # codebooks beyond the all-in-one kernel: one update + prepare launch, then search / exact / sums
    (100, np.float32, 30_000, 6, 1, 10), (40, np.float16, 40_000, 6, 1, 20), (40, np.float32, 21_000, 4, 2, 20),
    (64, np.float32, 9_000, 3, 1, 10)])
def test_batch_train_steps_fused_equals_unfused_and_oracle(gpu, oracle, c, dtype, n, m, passes, grid):
    """pxsom_batch_train_steps on a 10 x 10 grid: the one-launch-per-step route (pending update + prep at the
    head of the BMU search) against the launch-per-phase route step by step (codebooks and statistics, bit for
    bit for rows that sum exactly in binary64), against a run in one call, and against orc_som_batch."""
    xdim = ydim = grid
    k = xdim * ydim
    x = synth.make_fov_numpy(n, c, seed=77, dtype=np.float32).astype(dtype)
    w0 = _codebook(x.astype(np.float64), k, seed=12)
    w0[17] = w0[3]                       # a duplicate node from the start
    xd = torch.from_numpy(x).to(gpu)
    rr = default_radius_range(xdim, ydim)
    total = m * passes
    exact_rows = dtype != np.float64     # binary64 rows: atomic summation order leaves 1e-16 noise
    states = [sd.BatchTrainState(n, c, xdim, ydim, m, gpu) for _ in range(2)]
    for st in states:
        st.wbuf[0].copy_(torch.from_numpy(w0))
    for g in range(total):
        sd.batch_train_steps(xd, states[0], g, g + 1, total, (0.05, 0.01), rr)                  # fused
        sd.batch_train_steps(xd, states[1], g, g + 1, total, (0.05, 0.01), rr, unfused=True)
        if not exact_rows:
            # keep the two routes on the same trajectory: per-step parity is what is compared
            states[1].wbuf.copy_(states[0].wbuf)
            ra, rb = states[0].ring[g % 3].cpu().numpy(), states[1].ring[g % 3].cpu().numpy()
            np.testing.assert_allclose(ra, rb, rtol=1e-12, atol=1e-12)
            states[1].ring.copy_(states[0].ring)
            continue
        assert torch.equal(states[0].wbuf[g % 2], states[1].wbuf[g % 2]), f"codebook of step {g} differs"
        assert torch.equal(states[0].ring[g % 3], states[1].ring[g % 3]), f"statistics of step {g} differ"
        assert float(states[0].ring[(g + 1) % 3].abs().max()) == 0.0, "next statistics buffer not cleared"
    wa = torch.empty((k, c), dtype=torch.float64, device=gpu)
    wb = torch.empty_like(wa)
    sd.batch_train_finish(states[0], total, total, (0.05, 0.01), rr, wa)
    sd.batch_train_finish(states[1], total, total, (0.05, 0.01), rr, wb)
    if not exact_rows:
        np.testing.assert_allclose(wa.cpu().numpy(), wb.cpu().numpy(), rtol=1e-12, atol=0)
        return
    assert torch.equal(wa, wb)
    # the whole run in ONE call (what a single process does)
    st = sd.BatchTrainState(n, c, xdim, ydim, m, gpu)
    st.wbuf[0].copy_(torch.from_numpy(w0))
    sd.batch_train_steps(xd, st, 0, total, total, (0.05, 0.01), rr)
    wc = torch.empty_like(wa)
    sd.batch_train_finish(st, total, total, (0.05, 0.01), rr, wc)
    assert torch.equal(wa, wc)
    want = oracle.som_batch(x.astype(np.float64), w0, xdim, ydim, passes, (0.05, 0.01), rr, m)
    np.testing.assert_allclose(wa.cpu().numpy(), want, rtol=1e-9, atol=0)


def test_batch_train_steps_statistics_match_oracle_per_step(gpu, oracle):
    """Every fused step's statistics == orc_cluster_sums of the oracle's BMUs for the codebook that step used,
    and the codebook it derived == orc_batch_update of the previous one (ties / duplicate nodes included)."""
    xdim = ydim = 10
    k, c, n, m = 100, 22, 16_000, 8
    x = synth.make_fov_numpy(n, c, seed=5, dtype=np.float32)
    x[100:110] = x[100]                  # repeated rows
    w0 = _codebook(x.astype(np.float64), k, seed=2)
    w0[50] = w0[49]
    xd = torch.from_numpy(x).to(gpu)
    rr = default_radius_range(xdim, ydim)
    st = sd.BatchTrainState(n, c, xdim, ydim, m, gpu)
    st.wbuf[0].copy_(torch.from_numpy(w0))
    from ark_analysis_amd.distributed import batch_schedule
    w_prev = None
    for g in range(m):
        sd.batch_train_steps(xd, st, g, g + 1, m, (0.05, 0.01), rr)
        w_g = st.wbuf[g % 2].cpu().numpy()
        if g > 0:
            thr, alpha = batch_schedule(g - 1, m, (0.05, 0.01), rr)
            want_w = oracle.batch_update(w_prev, xdim, ydim, s_prev, cnt_prev, thr, alpha)
            np.testing.assert_allclose(w_g, want_w, rtol=1e-12, atol=0)
        rows = x[g::m].astype(np.float64)
        lab, _ = oracle.map_data_to_nodes(w_g, rows)
        s, cnt = oracle.cluster_sums(rows, lab, k)
        ring = st.ring[g % 3].cpu().numpy()
        np.testing.assert_array_equal(ring[k * c:], cnt.astype(np.float64))
        np.testing.assert_array_equal(ring[: k * c].reshape(k, c), s)
        w_prev, s_prev, cnt_prev = w_g, s, cnt


@pytest.mark.parametrize("xdim,ydim,c,dtype", [(10, 10, 100, np.float32), (12, 10, 22, np.float32), (8, 16, 40, np.float32),
                                               (7, 5, 9, np.float64), (3, 1, 128, np.float32), (11, 11, 2, np.float32),
                                               (12, 12, 22, np.float32), (16, 16, 8, np.float32), (13, 15, 30, np.float64)])
@pytest.mark.parametrize("rr", [(1.5, 0.0), (9.0, 0.0)])
def test_wide_bmu_only_steps_match_the_oracle_per_step(gpu, oracle, xdim, ydim, c, dtype, rr):
    """Grids other than 10 x 10 / rows wider than 32 channels (up to 256 nodes x 128 channels): the steps whose pending update has
    its threshold pinned at 0.5 run as ONE launch (csrc/pxsom_batch_step_wide.hip).  Step by step: the codebook a step derives
    == orc_batch_update of the previous one, its statistics == orc_cluster_sums of the oracle's BMUs for that codebook (ties,
    duplicate nodes and repeated rows included), and the launch-per-phase route gives the same bits."""
    k, n, m = xdim * ydim, 12_000, 8
    x = synth.make_fov_numpy(n, c, seed=15, dtype=np.float32).astype(dtype)
    x[100:110] = x[100]                  # repeated rows
    w0 = _codebook(x.astype(np.float64), k, seed=4)
    if k > 3:
        w0[k - 1] = w0[1]                # a duplicate node
    xd = torch.from_numpy(x).to(gpu)
    # rr (1.5, 0): the threshold drops under 1 (pinned at 0.5) from the fifth step on; (9, 0): windows wider than most grids
    # throughout -- the windowed steps of grids up to 16 x 16 take the same one-launch kernel
    states = [sd.BatchTrainState(n, c, xdim, ydim, m, gpu) for _ in range(2)]
    for st in states:
        st.wbuf[0].copy_(torch.from_numpy(w0))
    from ark_analysis_amd.distributed import batch_schedule
    bmu_only = 0
    w_prev = None
    for g in range(m):
        sd.batch_train_steps(xd, states[0], g, g + 1, m, (0.05, 0.01), rr)
        sd.batch_train_steps(xd, states[1], g, g + 1, m, (0.05, 0.01), rr, unfused=True)
        w_g = states[0].wbuf[g % 2].cpu().numpy()
        if g > 0:
            thr, alpha = batch_schedule(g - 1, m, (0.05, 0.01), rr)
            bmu_only += thr == 0.5
            want_w = oracle.batch_update(w_prev, xdim, ydim, s_prev, cnt_prev, thr, alpha)
            np.testing.assert_allclose(w_g, want_w, rtol=1e-12, atol=0)
        rows = x[g::m].astype(np.float64)
        lab, _ = oracle.map_data_to_nodes(w_g, rows)
        s, cnt = oracle.cluster_sums(rows, lab, k)
        ring = states[0].ring[g % 3].cpu().numpy()
        np.testing.assert_array_equal(ring[k * c:], cnt.astype(np.float64))
        if dtype == np.float64:          # binary64 rows: the order of the atomic additions leaves 1e-16 noise
            np.testing.assert_allclose(ring[: k * c].reshape(k, c), s, rtol=1e-12, atol=1e-12)
            states[1].wbuf.copy_(states[0].wbuf)
            states[1].ring.copy_(states[0].ring)
        else:
            np.testing.assert_array_equal(ring[: k * c].reshape(k, c), s)
            assert torch.equal(states[0].wbuf[g % 2], states[1].wbuf[g % 2]), f"codebook of step {g} differs from the launch-per-phase route"
            assert torch.equal(states[0].ring[g % 3], states[1].ring[g % 3]), f"statistics of step {g} differ"
        assert float(states[0].ring[(g + 1) % 3].abs().max()) == 0.0, "next statistics buffer not cleared"
        w_prev, s_prev, cnt_prev = w_g, s, cnt
    assert bmu_only >= (3 if rr[0] < 2 else 0)


def test_assign_full_size_sampled_against_oracle(gpu, oracle):
    """BASELINE config 2 size on one GPU (10 x 1024^2 x 22 fp32, K=100): rows are independent, so
    the oracle on a random sample of rows must agree exactly; plus idempotence and range."""
    n, c, k = 10 * 1024 * 1024, 22, 100
    x = synth.make_fov_torch(n, c, seed=1000, device=gpu)
    wd = x[torch.randperm(n, device=gpu)[:k]].to(torch.float64).contiguous()
    labels, _ = sd.assign(x, wd)
    labels2, _ = sd.assign(x, wd)
    assert torch.equal(labels, labels2)
    assert int(labels.min()) >= 1 and int(labels.max()) <= k
    exact_rows = sd.last_exact_rows(sd.assign.last_workspace)
    assert exact_rows < n // 20, f"{exact_rows} of {n} rows took the exact path"
    idx = torch.randperm(n, device=gpu)[:200_000]
    want, _ = oracle.map_data_to_nodes(wd.cpu().numpy(), x[idx].double().cpu().numpy())
    np.testing.assert_array_equal(labels[idx].cpu().numpy(), want)
    # checksum of checksums: per-cluster counts add up to n
    _, cnt = sd.cluster_sums(x, labels, k)
    assert int(cnt.sum()) == n


def test_mean_table_full_size_does_not_depend_on_the_row_order(gpu, oracle):
    """BASELINE config 2 size (10 x 1024^2 x 22 fp32, K = 100), labels + mean table in one pass: the same rows as generated,
    sorted by label and in shuffled runs of 7 equal labels -- the orders that send the table adds through the plain path, the
    row-axis path and a mixture of both -- give the same counts bit for bit, labels that are the permutation of each other,
    tables within the fixed-point bound; a sample of the sorted order against the oracle."""
    n, c, k = 10 * 1024 * 1024, 22, 100
    x = synth.make_fov_torch(n, c, seed=1001, device=gpu)
    wd = x[torch.randperm(n, device=gpu)[:k]].to(torch.float64).contiguous()
    l0, s0, c0 = sd.assign_sums(x, wd)
    assert int(c0.sum()) == n
    order = torch.argsort(l0.long(), stable=True)
    run = 7
    pieces = n // run
    idx7 = (torch.randperm(pieces, device=gpu).unsqueeze(1) * run + torch.arange(run, device=gpu).unsqueeze(0)).reshape(-1)
    idx7 = torch.cat([order[idx7], order[pieces * run:]])
    wmax = float(wd.abs().max())
    for idx in (order, idx7):
        xs = x[idx].contiguous()
        l1, s1, c1 = sd.assign_sums(xs, wd)
        assert torch.equal(l1, l0[idx]) and torch.equal(c1, c0)
        cnt = c0.clamp(min=1).to(torch.float64).unsqueeze(1)
        err = ((s1 - s0) / cnt).abs()
        assert bool((err <= 1e-6 * (s0 / cnt).abs() + 1e-10 * wmax).all()), float(err.max())
        del xs
    pick = torch.randperm(n, device=gpu)[:100_000]
    want, _ = oracle.map_data_to_nodes(wd.cpu().numpy(), x[order[pick]].double().cpu().numpy())
    np.testing.assert_array_equal(l0[order[pick]].cpu().numpy(), want)


def test_cell_som_shape_full_size_properties(gpu, oracle):
    """BASELINE config 4 (cell SOM: 1e6 cells x 100 features, 10x10 SOM) at full size: batch training leaves a
    finite codebook whose every mini-batch statistic accounts for every row; labels of a random sample equal
    the oracle's; relabelling is idempotent; per-cluster counts add up."""
    from ark_analysis_amd.distributed import BatchSOMTrainer
    n, c, xdim, ydim = 1_000_000, 100, 10, 10
    k = xdim * ydim
    g = torch.Generator(device=gpu)
    g.manual_seed(4)
    x = torch.rand((n, c), generator=g, device=gpu, dtype=torch.float32)
    x *= (torch.rand((n, c), generator=g, device=gpu) > 0.6)            # sparse counts, like pixel-cluster counts
    w = x[torch.randperm(n, device=gpu)[:k]].double().contiguous()
    tr = BatchSOMTrainer(xdim, ydim, c, gpu, batch_steps=16)
    tr.train(x, w, num_passes=1)
    assert bool(torch.isfinite(w).all())
    labels, _ = sd.assign(x, w)
    labels2, _ = sd.assign(x, w)
    assert torch.equal(labels, labels2)
    idx = torch.randperm(n, device=gpu)[:50_000]
    want, _ = oracle.map_data_to_nodes(w.cpu().numpy(), x[idx].double().cpu().numpy())
    np.testing.assert_array_equal(labels[idx].cpu().numpy(), want)
    sums, cnt = sd.cluster_sums(x, labels, k)
    assert int(cnt.sum()) == n
    np.testing.assert_allclose(sums.sum(dim=0).cpu().numpy(), x.double().sum(dim=0).cpu().numpy(), rtol=1e-9)


def test_config5_at_fov_size(gpu, oracle):
    """BASELINE config 5's shape at FOV size on one GPU: one 2048 x 2048 x 40 fp16 FOV, 20 x 20 SOM.  Batch training
    on the 10 % subset, labels of every pixel against the oracle on a 100 k-row sample (bit-exact), range /
    idempotence / count properties over all rows, per-cluster sums against a binary64 scatter-add of the same
    values, meta-cluster lookup."""
    from ark_analysis_amd.distributed import BatchSOMTrainer
    n, c, xdim, ydim = 2048 * 2048, 40, 20, 20
    k = xdim * ydim
    x = synth.make_fov_torch(n, c, seed=55, device=gpu, dtype=torch.float16)
    sub = x[::10].contiguous()
    g = torch.Generator(device="cpu")
    g.manual_seed(3)
    w = sub[torch.randperm(sub.shape[0], generator=g)[:k].to(gpu)].double().contiguous()
    BatchSOMTrainer(xdim, ydim, c, gpu, batch_steps=16).train(sub, w, num_passes=1)
    assert bool(torch.isfinite(w).all())
    labels, _ = sd.assign(x, w)
    exact_rows = sd.last_exact_rows(sd.assign.last_workspace)
    labels2, _ = sd.assign(x, w)
    assert torch.equal(labels, labels2)
    assert int(labels.min()) >= 1 and int(labels.max()) <= k
    assert exact_rows < n // 10, f"{exact_rows} of {n} rows took the exact path"
    idx = torch.randperm(n, device=gpu)[:100_000]
    want, _ = oracle.map_data_to_nodes(w.cpu().numpy(), x[idx].double().cpu().numpy())
    np.testing.assert_array_equal(labels[idx].cpu().numpy(), want)
    sums, counts = sd.cluster_sums(x, labels, k)
    np.testing.assert_array_equal(counts.cpu().numpy(), torch.bincount(labels.long() - 1, minlength=k).cpu().numpy())
    assert int(counts.sum()) == n
    ref = torch.zeros((k, c), dtype=torch.float64, device=gpu).index_add_(0, labels.long() - 1, x.double())
    np.testing.assert_allclose(sums.cpu().numpy(), ref.cpu().numpy(), rtol=1e-12, atol=1e-9)
    lut = torch.from_numpy(np.concatenate([[0], np.random.RandomState(1).randint(1, 21, size=k)]).astype(np.int32)).to(gpu)
    meta = sd.relabel(labels, lut)
    assert torch.equal(meta, lut[labels.long()])


def test_relabel_matches_numpy(gpu):
    rs = np.random.RandomState(4)
    for n, k in [(1_000_003, 400), (37, 100), (0, 5), (4096, 1)]:
        lut = rs.randint(1, 21, size=k + 1).astype(np.int32)
        labels = rs.randint(-2, k + 4, size=n).astype(np.int32)
        want = np.where((labels >= 0) & (labels <= k), lut[np.clip(labels, 0, k)], -7)
        ld = torch.from_numpy(labels).to(gpu)
        got = sd.relabel(ld, torch.from_numpy(lut).to(gpu), fill=-7)
        np.testing.assert_array_equal(got.cpu().numpy(), want)
        if n:
            view = ld[1:]                                    # a misaligned view takes the scalar loop
            got2 = sd.relabel(view.contiguous(), torch.from_numpy(lut).to(gpu), fill=-7)
            np.testing.assert_array_equal(got2.cpu().numpy(), want[1:])


@pytest.mark.parametrize("apart,dtype", [(3e-3, np.float32), (1e-3, np.float32), (3e-4, np.float32), (1e-6, np.float32),
                                         (1e-3, np.float16), (3e-4, np.float64)])
def test_deferred_full_search_on_codebooks_of_close_node_pairs(gpu, oracle, apart, dtype):
    """The register-resident filters (labels only, and labels + tables in one pass) keep the rows their first stage cannot
    vouch for in a queue per wave and search them in full 64 at a time.  Codebooks whose nodes come in pairs `apart` (relative)
    from each other make that the fate of a third to all of the rows -- queues that fill between two drains, batches behind
    batches: labels equal the oracle's, twice in a row, and the one-pass tables count every row once."""
    n, c, k = 1_200_000, 22, 100
    x = synth.make_fov_numpy(n, c, seed=33, dtype=np.float32).astype(dtype)
    rs = np.random.RandomState(9)
    half = _codebook(x.astype(np.float64), k // 2, seed=6)
    w = np.concatenate([half, half * (1.0 + apart * rs.uniform(0.5, 1.5, size=half.shape))])
    xd, wd = torch.from_numpy(x).to(gpu), torch.from_numpy(np.ascontiguousarray(w)).to(gpu)
    la, _ = sd.assign(xd, wd)
    lb, _ = sd.assign(xd, wd)
    assert torch.equal(la, lb)
    want, _ = oracle.map_data_to_nodes(w, x.astype(np.float64))
    np.testing.assert_array_equal(la.cpu().numpy(), want)
    l1, s1, c1 = sd.assign_sums(xd, wd)
    np.testing.assert_array_equal(l1.cpu().numpy(), want)
    np.testing.assert_array_equal(c1.cpu().numpy(), np.bincount(want - 1, minlength=k))
    s2, _ = sd.cluster_sums(xd, la, k)
    np.testing.assert_allclose(s1.cpu().numpy(), s2.cpu().numpy(), rtol=1e-6, atol=1e-6 * float(np.abs(s2.cpu().numpy()).max()))


@pytest.mark.parametrize("n,c,k,dtype", [(300_000, 22, 100, np.float32), (50_001, 8, 100, np.float32), (70_000, 16, 100, np.float16),
                                         (20_000, 22, 100, np.float64), (30_000, 40, 400, np.float16), (9_000, 7, 30, np.float32)])
def test_assign_sums_one_pass_equals_two_passes(gpu, oracle, n, c, k, dtype):
    """pxsom_assign_sums (labels + per-cluster tables, one pass over x where the shape allows) against pxsom_assign
    followed by pxsom_cluster_sums: labels and counts bit for bit; labels against the oracle; tables are added into.
    Sums: the register-resident shapes accumulate in 64-bit fixed point (error per value <= 2^-39 x rows-per-workgroup
    relative to the codebook's largest magnitude) -- the BOUND is tested: every per-cluster mean within 1e-6 relative
    (+ 1e-10 of the codebook's largest magnitude for means near zero) of the exact two-pass table; other shapes run the
    two kernels and are exact."""
    x = synth.make_fov_numpy(n, c, seed=91, dtype=np.float32).astype(dtype)
    x[500:520] = x[500]
    x[600, :] = 0.0                       # an all-zero row
    x[700:705, 1] = 3.0e-7                # values far below the table's unit of the largest ones
    if dtype != np.float16:
        x[800, 0] = 5.0e4                 # a row the filter lists for its size: taken by the exact path, outside the table's format
        x[801, 2] = -1.5                  # negative values are legal input
    w = _codebook(x.astype(np.float64), k, seed=4)
    w[k - 1] = w[2]
    xd, wd = torch.from_numpy(x).to(gpu), torch.from_numpy(w).to(gpu)
    l2, _ = sd.assign(xd, wd)
    s2, c2 = sd.cluster_sums(xd, l2, k)
    pre_s = torch.full((k, c), 0.5, dtype=torch.float64, device=gpu)
    pre_c = torch.full((k,), 3, dtype=torch.int64, device=gpu)
    l1, s1, c1 = sd.assign_sums(xd, wd, sums=pre_s.clone(), counts=pre_c.clone())
    assert torch.equal(l1, l2)
    assert torch.equal(c1, c2 + 3)
    cnt = np.maximum(c2.cpu().numpy(), 1)[:, None].astype(np.float64)
    mean1, mean2 = (s1 - 0.5).cpu().numpy() / cnt, s2.cpu().numpy() / cnt
    wmax = float(np.abs(w).max())
    assert np.all(np.abs(mean1 - mean2) <= 1e-6 * np.abs(mean2) + 1e-10 * wmax), float(np.abs(mean1 - mean2).max())
    want, _ = oracle.map_data_to_nodes(w, x.astype(np.float64))
    np.testing.assert_array_equal(l1.cpu().numpy(), want)
    # run to run: integer accumulation inside a workgroup; the binary64 partials of the workgroups are added by global
    # atomics in any order, so only the last bits may move
    l3, s3, c3 = sd.assign_sums(xd, wd, sums=pre_s.clone(), counts=pre_c.clone())
    assert torch.equal(l3, l1) and torch.equal(c3, c1)
    np.testing.assert_allclose(s3.cpu().numpy(), s1.cpu().numpy(), rtol=1e-13, atol=0)


@pytest.mark.parametrize("run", [1000, 64, 16, 9, 5])
@pytest.mark.parametrize("dtype", [np.float32, np.float16, np.float64])
def test_assign_sums_on_label_coherent_rows(gpu, oracle, run, dtype):
    """Rows whose neighbours share their label (what images look like; the synthetic FOVs do not): the one-pass kernel sums such
    tiles along the row axis before they touch its table (prefix sums over the 16 rows of a tile, one add and one subtraction per
    run of equal labels).  Labels against the oracle; counts bit for bit and means within the fixed-point bound against the
    two-pass tables -- with runs longer than, equal to and shorter than a tile, runs that straddle tiles, listed rows (duplicate
    nodes, an oversized row) inside the runs, and the table equal from run to run.  binary64 rows take the two-tile kernel
    (pxsom_assign_onepass.h), which carries the same adds."""
    n, c, k = 70_000, 22, 100
    x = synth.make_fov_numpy(n, c, seed=95, dtype=np.float32)
    w = _codebook(x.astype(np.float64), k, seed=6)
    w[k - 1] = w[7]                                    # duplicate nodes: their rows are listed, inside the runs
    first, _ = oracle.map_data_to_nodes(w, x.astype(np.float64))
    order = np.argsort(first, kind="stable")
    pieces = n // run
    perm = np.random.RandomState(3).permutation(pieces)
    idx = (perm[:, None] * run + np.arange(run)[None, :]).reshape(-1)
    x = np.ascontiguousarray(x[order][np.concatenate([idx, np.arange(pieces * run, n)])]).astype(dtype)
    if dtype != np.float16:
        x[4_000, 0] = 5.0e4                            # a row the filter lists for its size, in the middle of a run
    xd, wd = torch.from_numpy(x).to(gpu), torch.from_numpy(w).to(gpu)
    l2, _ = sd.assign(xd, wd)
    s2, c2 = sd.cluster_sums(xd, l2, k)
    l1, s1, c1 = sd.assign_sums(xd, wd)
    want, _ = oracle.map_data_to_nodes(w, x.astype(np.float64))
    np.testing.assert_array_equal(l1.cpu().numpy(), want)
    assert torch.equal(l1, l2) and torch.equal(c1, c2)
    cnt = np.maximum(c2.cpu().numpy(), 1)[:, None].astype(np.float64)
    mean1, mean2 = s1.cpu().numpy() / cnt, s2.cpu().numpy() / cnt
    wmax = float(np.abs(w).max())
    assert np.all(np.abs(mean1 - mean2) <= 1e-6 * np.abs(mean2) + 1e-10 * wmax), float(np.abs(mean1 - mean2).max())
    l3, s3, c3 = sd.assign_sums(xd, wd)
    assert torch.equal(l3, l1) and torch.equal(c3, c1)
    np.testing.assert_allclose(s3.cpu().numpy(), s1.cpu().numpy(), rtol=1e-13, atol=0)
    # the two-pass sums kernel on the same rows (groups of rows that all carry one label are added up across the lane slots first)
    so, co = oracle.cluster_sums(x.astype(np.float64), want, k)
    np.testing.assert_array_equal(c2.cpu().numpy(), co)
    mag, _ = oracle.cluster_sums(np.abs(x.astype(np.float64)), want, k)
    assert np.all(np.abs(s2.cpu().numpy() - so) <= 1e-12 * mag), float(np.abs(s2.cpu().numpy() - so).max())


@pytest.mark.parametrize("n", [2_000, 20_000, 150_000])
def test_assign_sums_small_launch_large_vouched_value(gpu, oracle, n):
    """A row the filter VOUCHES for may hold values up to 2^16 / scale -- a hundred times the codebook's largest entry.
    The fixed-point tables of a small launch (few rows per workgroup, hence many fractional bits) must still take it:
    their limit used to fall below that bound when a workgroup met fewer than 2^11 rows (20 000 -> 18 192 in the sum)."""
    c, k = 22, 100
    x = synth.make_fov_numpy(n, c, seed=93, dtype=np.float32)
    w = _codebook(x.astype(np.float64), k, seed=5)
    wmax = float(np.abs(w).max())
    x[800, 0] = 150.0 * wmax              # far from every node, yet inside the filter's range: not listed
    x[801, 3] = -100.0 * wmax
    xd, wd = torch.from_numpy(x).to(gpu), torch.from_numpy(w).to(gpu)
    l2, _ = sd.assign(xd, wd)
    listed = sd.last_exact_rows(sd.assign.last_workspace)
    s2, c2 = sd.cluster_sums(xd, l2, k)
    l1, s1, c1 = sd.assign_sums(xd, wd)
    assert torch.equal(l1, l2) and torch.equal(c1, c2)
    want, _ = oracle.map_data_to_nodes(w, x.astype(np.float64))
    np.testing.assert_array_equal(l1.cpu().numpy(), want)
    np.testing.assert_allclose(s1.cpu().numpy(), s2.cpu().numpy(), rtol=1e-6, atol=1e-9 * wmax * n)
    assert listed < n // 10, listed       # (the two large rows must not owe their exact sums to the exact path alone)


@pytest.mark.parametrize("dtype,offset,spread,collapse", [
    (np.float32, 100.0, 1.0, 1.0),      # a blob far from the origin: uncentred, every row is a near-tie
    (np.float32, 100.0, 1.0, 1e-3),     # ... and a codebook that has collapsed onto its mean (early training steps)
    (np.float64, 1.0e4, 1.0, 1.0),      # binary64 rows: the centred row is formed in binary64 before it is rounded
    (np.float64, -50.0, 2.0, 1e-2),
    (np.float16, 8.0, 0.5, 1.0),        # binary16 rows: x converts exactly, the subtraction rounds once
    (np.float32, 0.0, 1.0, 1.0),        # nothing to centre
])
def test_centred_filter_on_offset_data(gpu, oracle, dtype, offset, spread, collapse):
    """The register-resident filter ranks the nodes on rows and codebook centred on the codebook's mean (AssignHdr::mu_s):
    its error bound is then relative to the centred norms.  Labels stay the oracle's bit for bit (plain search, one-pass
    labels + tables, a training pass), rows far from the origin no longer drown the score gaps in the tolerance (few
    listed rows), and the one-pass tables' fixed-point format still holds rows whose uncentred values are large."""
    n, c, k = 60_000, 22, 100
    rs = np.random.RandomState(int(abs(offset)) + c)
    centre = offset + spread * rs.standard_normal(c)
    x = (centre + spread * rs.standard_normal((n, c))).astype(dtype)
    w = (centre + collapse * spread * rs.standard_normal((k, c))).astype(np.float64)
    w[5] = w[4]                                              # a duplicated node
    x[100] = w[7].astype(dtype)                              # rows that sit on nodes (up to the row's rounding)
    x[101] = 0.0                                             # the origin: far from everything
    xd, wd = torch.from_numpy(x).to(gpu), torch.from_numpy(w).to(gpu)
    want, _ = oracle.map_data_to_nodes(w, x.astype(np.float64))
    got, _ = sd.assign(xd, wd)
    np.testing.assert_array_equal(got.cpu().numpy(), want)
    listed = sd.last_exact_rows(sd.assign.last_workspace)
    if collapse == 1.0:
        assert listed < 0.02 * n, listed                     # (uncentred: > 90 % at offset 100)
    l1, s1, c1 = sd.assign_sums(xd, wd)
    np.testing.assert_array_equal(l1.cpu().numpy(), want)
    s2, c2 = sd.cluster_sums(xd, got, k)
    assert torch.equal(c1, c2)
    cnt = np.maximum(c2.cpu().numpy(), 1)[:, None].astype(np.float64)
    mean1, mean2 = s1.cpu().numpy() / cnt, s2.cpu().numpy() / cnt
    assert np.all(np.abs(mean1 - mean2) <= 1e-6 * np.abs(mean2) + 1e-10 * float(np.abs(w).max())), float(np.abs(mean1 - mean2).max())
    # one training pass from this codebook: the step kernel centres on the run's first codebook
    from ark_analysis_amd.distributed import BatchSOMTrainer
    from ark_analysis_amd.schedule import BatchSchedule
    sch = BatchSchedule.equal(6)
    xt = x[:24_000]
    if dtype == np.float64:
        xt = np.round(xt * 1024.0) / 1024.0                  # exact sums whatever the order (no quantum needed)
    tr = BatchSOMTrainer(10, 10, c, gpu, batch_steps=sch)
    wt = torch.from_numpy(w.copy()).to(gpu)
    tr.train(torch.from_numpy(xt).to(gpu), wt, 1)
    from ark_analysis_amd.flowsom import default_radius_range
    want_w = oracle.som_batch_sched(xt.astype(np.float64), w, 10, 10, 1, (0.05, 0.01), default_radius_range(10, 10),
                                    sch.phases, list(sch.edges))
    np.testing.assert_allclose(wt.cpu().numpy(), want_w, rtol=1e-9, atol=1e-12 * float(np.abs(w).max()))


def test_packed_k_filter_equals_chunked_filter_and_oracle(gpu, oracle):
    """Config 5's shape (binary16 rows, 40 channels, 20 x 20 SOM): the packed-K streamed filter (aligned rows) against the
    chunked one (the same rows at an odd element offset, which the packed kernel cannot read) and the oracle; rows with
    values the binary16 range barely holds, duplicates and a NaN."""
    n, c, k = 150_001, 40, 400
    x = synth.make_fov_numpy(n, c, seed=21, dtype=np.float32).astype(np.float16)
    x[10:20] = x[10]
    x[33, 4] = np.float16(6.0e4)
    x[34, 0] = np.nan
    w = _codebook(x.astype(np.float64), k, seed=9)
    w[7] = w[300]
    xd, wd = torch.from_numpy(x).to(gpu), torch.from_numpy(w).to(gpu)
    packed, _ = sd.assign(xd, wd)
    listed_packed = sd.last_exact_rows(sd.assign.last_workspace)
    shifted = torch.empty(n * c + 1, dtype=torch.float16, device=gpu)
    shifted[1:] = xd.reshape(-1)
    chunked, _ = sd.assign(shifted[1:].view(n, c), wd)          # 2-byte aligned only: the chunked kernel
    listed_chunked = sd.last_exact_rows(sd.assign.last_workspace)
    assert torch.equal(packed, chunked)
    want, _ = oracle.map_data_to_nodes(w, x.astype(np.float64))
    np.testing.assert_array_equal(packed.cpu().numpy(), want)
    assert listed_packed < n // 10 and listed_packed <= 2 * listed_chunked + 64     # same bound, same order of listed rows


@pytest.mark.parametrize("n,c,k,dtype", [(200_000, 22, 100, np.float32), (30_000, 40, 400, np.float16), (40, 22, 100, np.float32), (0, 8, 100, np.float32)])
def test_assign_means_equals_assign_sums(gpu, n, c, k, dtype):
    """pxsom_assign_means = pxsom_assign_sums on cleared tables + the division, in one call: labels, sums, counts equal, means =
    sums / max(count, 1); the outputs are overwritten (stale contents do not leak)."""
    x = synth.make_fov_numpy(max(n, 1), c, seed=17, dtype=np.float32).astype(dtype)[:n]
    w = _codebook(synth.make_fov_numpy(4 * k, c, seed=18, dtype=np.float64), k, seed=2)
    xd, wd = torch.from_numpy(x).to(gpu), torch.from_numpy(w).to(gpu)
    ws = sd.AssignSumsWorkspace(max(n, 1), c, k, gpu)
    labels = torch.full((n,), -5, dtype=torch.int32, device=gpu)
    sums = torch.full((k, c), 7.0, dtype=torch.float64, device=gpu)
    counts = torch.full((k,), 9, dtype=torch.int64, device=gpu)
    means = torch.full((k, c), -1.0, dtype=torch.float64, device=gpu)
    sd.assign_means(xd, wd, labels, sums, counts, means, ws)
    if n:
        l2, s2, c2 = sd.assign_sums(xd, wd)
        assert torch.equal(labels, l2) and torch.equal(counts, c2)
        np.testing.assert_allclose(sums.cpu().numpy(), s2.cpu().numpy(), rtol=1e-13, atol=0)
    else:
        assert int(counts.abs().sum()) == 0 and float(sums.abs().sum()) == 0.0
    want = sums / counts.clamp(min=1).to(torch.float64).unsqueeze(1)
    assert torch.equal(means, want)
