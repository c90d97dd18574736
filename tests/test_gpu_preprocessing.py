"""Parity of the pre-processing HIP kernels with fixtures produced by the reference's own numerics
(scipy.ndimage.gaussian_filter, create_fov_pixel_data, pandas / numpy quantiles) and with the
oracle.  `-m gpu` only."""
import os

import numpy as np
import pandas as pd
import pytest
import torch

from ark_analysis_amd import som_device as sd

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_gaussian_blur_matches_scipy_fixture_and_oracle(gpu, oracle, tag):
    g = np.load(os.path.join(GOLD, f"g2_fovpixel_{tag}.npz"))
    img = torch.from_numpy(g["img"].copy()).to(gpu)
    sd.gaussian_blur_hwc(img, 2.0)
    got = img.cpu().numpy()
    np.testing.assert_allclose(got, g["blurred"], rtol=1e-13, atol=1e-300)        # scipy's output
    np.testing.assert_array_equal(got, oracle.gaussian_blur_hwc(g["img"], 2.0))   # same op order: bit-equal


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_float32_create_fov_pixel_data_matches_reference(gpu, oracle, tag):
    """float32 image in -> the reference's float32 tables out (blur passes stored as float32, binary32 row
    sums and divisions), through the kernels and through the create_fov_pixel_data mirror."""
    from ark_analysis_amd.phenotyping import pixie_preprocessing
    g = np.load(os.path.join(GOLD, f"g2_fovpixel_{tag}_f32.npz"))
    h, w, c = g["img"].shape
    img = torch.from_numpy(g["img"].astype(np.float64)).to(gpu)
    sd.gaussian_blur_hwc(img, 2.0, f32_semantics=True)
    np.testing.assert_array_equal(img.cpu().numpy().astype(np.float32), g["blurred"])
    np.testing.assert_array_equal(img.cpu().numpy(), oracle.gaussian_blur_hwc(g["img"], 2.0, f32=True))
    rows, kept = sd.rowsum_filter_normalize(img.view(h * w, c), float(g["thresh"]), f32_semantics=True)
    np.testing.assert_array_equal(kept.cpu().numpy(), g["kept_index"])
    np.testing.assert_array_equal(rows.cpu().numpy().astype(np.float32), g["rows"])
    chans = ["chan%d" % i for i in range(c)]
    np.random.seed(7)
    full, sub = pixie_preprocessing.create_fov_pixel_data("fov0", list(chans), g["img"].copy(), None,
                                                          pixel_thresh_val=g["thresh"])
    assert full[chans].values.dtype == np.float32
    np.testing.assert_array_equal(full[chans].values, g["rows"])
    np.testing.assert_array_equal((full["row_index"].values * w + full["column_index"].values), g["kept_index"])
    np.testing.assert_array_equal(sub.index.values, g["subset_index"])          # same seeded sample


def test_gaussian_blur_small_and_odd_shapes(gpu, oracle):
    rs = np.random.RandomState(0)
    for (h, w, c, sigma) in [(1, 1, 1, 2.0), (3, 50, 2, 2.0), (40, 5, 3, 1.0), (9, 9, 4, 3.0), (64, 64, 22, 2.0)]:
        img = rs.rand(h, w, c)
        t = torch.from_numpy(img.copy()).to(gpu)
        sd.gaussian_blur_hwc(t, sigma)
        np.testing.assert_array_equal(t.cpu().numpy(), oracle.gaussian_blur_hwc(img, sigma))


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_rowsum_filter_normalize_matches_reference_fixture(gpu, tag):
    g = np.load(os.path.join(GOLD, f"g2_fovpixel_{tag}.npz"))
    h, w, c = g["img"].shape
    x = torch.from_numpy(g["blurred"].reshape(-1, c).copy()).to(gpu)
    rows, kept = sd.rowsum_filter_normalize(x, float(g["thresh"]))
    np.testing.assert_array_equal(kept.cpu().numpy(), g["kept_index"])
    np.testing.assert_array_equal(rows.cpu().numpy(), g["rows"])


def test_rowsum_filter_normalize_edge_cases(gpu, oracle):
    rs = np.random.RandomState(1)
    x = rs.gamma(0.5, 1.0, size=(70_001, 22))
    x[rs.rand(70_001) < 0.3] = 0.0                      # all-zero rows
    for thresh in (0.0, 3.0, 1e9):                      # keep most / some / none
        rows, kept = sd.rowsum_filter_normalize(torch.from_numpy(x).to(gpu), thresh)
        wr, wk = oracle.rowsum_filter_normalize(x, thresh, sum_mode=0)
        np.testing.assert_array_equal(kept.cpu().numpy(), wk)
        np.testing.assert_array_equal(rows.cpu().numpy(), wr)
    rows, kept = sd.rowsum_filter_normalize(torch.empty((0, 5), dtype=torch.float64, device=gpu), 0.0)
    assert rows.shape == (0, 5) and kept.numel() == 0


def test_create_fov_pixel_data_matches_reference_fixture(gpu):
    """The drop-in function end to end (blur + filter + normalise on the GPU, DataFrame on the host)."""
    from ark_analysis_amd.phenotyping import pixie_preprocessing
    for tag in ("a", "c"):
        g = np.load(os.path.join(GOLD, f"g2_fovpixel_{tag}.npz"))
        h, w, c = g["img"].shape
        chans = ["chan%d" % i for i in range(c)]
        np.random.seed(7)
        full, sub = pixie_preprocessing.create_fov_pixel_data("fov0", list(chans), g["img"].copy(), None,
                                                              pixel_thresh_val=float(g["thresh"]))
        assert list(full.columns) == chans + ["fov", "row_index", "column_index"]
        np.testing.assert_array_equal(full["row_index"].values * w + full["column_index"].values, g["kept_index"])
        np.testing.assert_allclose(full[chans].values, g["rows"], rtol=1e-12, atol=0)
        assert np.allclose(full[chans].values.sum(axis=1), 1.0)
        assert len(sub) == int(g["subset_len"])


def test_normalize_columns_matches_reference_fixture(gpu):
    g = np.load(os.path.join(GOLD, "g1_normalize.npz"))
    out = sd.normalize_columns(torch.from_numpy(g["x"]).to(gpu), torch.from_numpy(g["norm"]).to(gpu))
    np.testing.assert_array_equal(out.cpu().numpy(), g["out"])


def test_quantile_matches_pandas_numpy_fixture(gpu):
    g = np.load(os.path.join(GOLD, "g3_quantiles.npz"))
    for i in range(8):
        x = torch.from_numpy(g["x%d" % i].reshape(-1, 1).copy()).to(gpu)
        for key, q, mode in (("q999_%d", (0.999 * 100) / 100, 0), ("q99pos_%d", 0.99, 1)):
            want = float(g[key % i])
            got = float(sd.quantile_nonzero(x, q, keep_mode=mode).cpu()[0])
            assert (np.isnan(got) and np.isnan(want)) or got == want, (key % i, got, want)


def test_quantile_many_columns_against_pandas(gpu):
    """The cohort statistic of create_pixel_matrix: per-channel 99.9 % of the non-zero values."""
    rs = np.random.RandomState(3)
    x = rs.gamma(0.4, 1.0, size=(200_003, 22))
    x[rs.rand(*x.shape) < 0.25] = 0.0
    x[:, 5] = 0.0                       # all-zero column -> NaN
    x[:, 6] = np.round(x[:, 6], 1)      # heavy duplicates
    x[:1000, 7] *= -1                   # negative values sort below the positive ones
    got = sd.quantile_nonzero(torch.from_numpy(x).to(gpu), (0.999 * 100) / 100, keep_mode=0).cpu().numpy()
    want = pd.DataFrame(x).replace(0, np.nan).quantile(q=0.999, axis=0).values
    assert np.isnan(got[5]) and np.isnan(want[5])
    ok = ~np.isnan(want)
    np.testing.assert_array_equal(got[ok], want[ok])
    # strided view (one FOV's slice of a bigger matrix) and another q
    got2 = sd.quantile_nonzero(torch.from_numpy(x).to(gpu)[:, 3:9], 0.05, keep_mode=1).cpu().numpy()
    for j in range(6):
        col = x[:, 3 + j]
        pos = col[col > 0]
        if len(pos):
            assert got2[j] == np.quantile(pos, 0.05)
        else:
            assert np.isnan(got2[j])


def test_float32_quantiles_match_numpy_bit_for_bit(gpu):
    """np.quantile on float32 data (index, fraction and interpolation in binary32): the TIFF-side
    percentiles of calculate_channel_percentiles / calculate_pixel_intensity_percentile."""
    from ark_analysis_amd import flowsom
    rs = np.random.RandomState(3)
    for trial in range(40):
        n = int(rs.choice([1, 2, 3, 17, 1000, 4097, 65536, 300_001]))
        img = rs.gamma(0.6, 2.0, size=n).astype(np.float32)
        img[rs.uniform(size=n) < 0.3] = 0
        if trial % 7 == 0:
            img = np.round(img)                               # many repeated values
        q = float(rs.choice([0.99, 0.999, 0.05, 0.5, 0.123456]))
        kept = img[img > 0]
        got = flowsom.positive_quantile_f32(img, q)
        if kept.size == 0:
            assert np.isnan(got)
        else:
            want = np.quantile(kept, q)
            assert want.dtype == np.float32 and got.dtype == np.float32
            assert got == want, (n, q, got, want)
    # integer images interpolate in binary64
    img16 = rs.randint(0, 4000, size=5000).astype(np.uint16)
    assert flowsom.positive_quantile_f32(img16, 0.99) == np.quantile(img16[img16 > 0], 0.99)


@pytest.mark.parametrize("c", [1, 5, 8, 9, 22, 23, 40, 128])
def test_total_intensity_quantile_matches_numpy(gpu, c):
    """np.quantile(np.sum(img / norm, axis=-1), 0.05) incl. numpy's float32 summation order."""
    from ark_analysis_amd import flowsom, som_device as sd
    rs = np.random.RandomState(c)
    img = rs.gamma(0.5, 1.5, size=(61, 47, c)).astype(np.float32)
    img[rs.uniform(size=img.shape) < 0.4] = 0
    norm = rs.uniform(0.5, 3.0, size=c).astype(np.float32)
    want_sum = np.sum(img / norm.reshape([1, 1, c]), axis=-1)
    got_sum = sd.scaled_rowsum_f32(torch.from_numpy(img.reshape(-1, c)).to(gpu), torch.from_numpy(norm).to(gpu))
    np.testing.assert_array_equal(got_sum.cpu().numpy().reshape(61, 47), want_sum)
    for q in (0.05, 0.5):
        assert flowsom.total_intensity_quantile_f32(img, norm, q) == np.quantile(want_sum, q)
    # every other dtype pairing: numpy promotes the division to binary64 (uint16 / int16 TIFF exports, float64
    # channel percentiles) -- same summation order, binary64 quantile
    for image, divisors in [(img.astype(np.float64), norm), (img, norm.astype(np.float64)),
                            ((img * 500).astype(np.uint16), norm.astype(np.float64) * 500),
                            ((img * 500).astype(np.int16), norm.astype(np.float64) * 500)]:
        want64 = np.sum(image / divisors.reshape([1, 1, c]), axis=-1)
        assert want64.dtype == np.float64
        got64 = sd.scaled_rowsum(torch.from_numpy(image.reshape(-1, c).astype(np.float64)).to(gpu),
                                 torch.from_numpy(divisors.astype(np.float64)).to(gpu))
        np.testing.assert_array_equal(got64.cpu().numpy().reshape(61, 47), want64)
        for q in (0.05, 0.5):
            assert flowsom.total_intensity_quantile_f32(image, divisors, q) == np.quantile(want64, q)


@pytest.mark.parametrize("h,w,c,f32", [(1024, 1024, 22, False), (300, 257, 7, True), (17, 9, 40, False), (130, 2000, 3, True),
                                       (65, 33, 128, False)])
def test_fast_blur_forms_equal_the_generic_form(gpu, oracle, h, w, c, f32):
    """The pipeline's radius-8 blur (register window down the rows, LDS tile along the columns, tiles dealt to the XCDs
    in contiguous runs) against the thread-per-output form on the same image: bit for bit, strip / tile / image borders
    included; a corner of the big image against the oracle."""
    g = torch.Generator(device=gpu)
    g.manual_seed(h * 7 + w)
    img = torch.empty((h, w, c), dtype=torch.float64, device=gpu).exponential_(1.0, generator=g)
    img.mul_((torch.rand((h, w, c), generator=g, device=gpu) >= 0.4).to(torch.float64))
    if f32:
        img = img.float().double()
    fast = sd.gaussian_blur_hwc(img.clone(), 2.0, f32_semantics=f32)
    slow = sd.gaussian_blur_hwc(img.clone(), 2.0, f32_semantics=f32, generic_form=True)
    assert torch.equal(fast, slow)
    hh, ww = min(h, 40), min(w, 48)          # the top-left corner depends on the top-left (hh + 8) x (ww + 8) pixels only
    if h >= hh + 8 and w >= ww + 8:
        want = oracle.gaussian_blur_hwc(img[:hh + 8, :ww + 8].cpu().numpy().copy(), 2.0, f32=f32)
        # (the cropped image reflects at its own lower / right border: compare the part no reflection reaches)
        np.testing.assert_array_equal(fast[:hh, :ww].cpu().numpy(), want[:hh, :ww])


@pytest.mark.parametrize("n,c,f32", [(1024 * 1024, 22, False), (100_003, 22, True), (5_000, 72, False), (3_001, 100, False),
                                     (257, 1, False), (70_000, 31, True)])
def test_rowfilter_and_quantiles_at_size(gpu, oracle, n, c, f32):
    """Row filter + normalisation (rows staged in LDS, coalesced both ways) and the one-sweep quantile kernels on
    inputs large enough for many workgroups: kept rows, values and per-column 99.9 % values against the oracle."""
    rs = np.random.RandomState(n % 1000 + c)
    x = rs.gamma(0.7, 1.0, size=(n, c))
    x[rs.rand(n, c) < 0.45] = 0.0
    x[::97] = 0.0                                        # all-zero rows are dropped
    if f32:
        x = x.astype(np.float32).astype(np.float64)
    xd = torch.from_numpy(x).to(gpu)
    rows, kept = sd.rowsum_filter_normalize(xd, 0.8, f32_semantics=f32)
    want_rows, want_kept = oracle.rowsum_filter_normalize(x, 0.8, sum_mode=2 if f32 else 0)
    np.testing.assert_array_equal(kept.cpu().numpy(), want_kept)
    np.testing.assert_array_equal(rows.cpu().numpy(), want_rows)
    q = sd.quantile_nonzero(rows, 0.999).cpu().numpy()
    cols = range(c) if c <= 8 else rs.choice(c, 6, replace=False)
    for j in cols:
        assert q[j] == oracle.quantile_nonzero(np.ascontiguousarray(want_rows[:, j]), 0.999, 0) or \
            (np.isnan(q[j]) and not np.any(want_rows[:, j] != 0))
