"""Randomised parity sweep: the HIP path against the CPU oracle on shapes, strides, dtypes and value patterns drawn
at random (seeded), far off the handful of shapes the other tests name.  Every case asserts the same bars as the
named tests: BMU labels bit-equal, per-cluster sums / counts equal, batch-rule updates and whole runs on exact-sum tables bit-equal, online codebook
bit-equal.  ``PXSOM_FUZZ_CASES`` sets the number of cases per test (default 12: seconds; the round's sweep ran 1500)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = int(os.environ.get("PXSOM_FUZZ_CASES", "12"))
SEED = int(os.environ.get("PXSOM_FUZZ_SEED", "20260928"))
TORCH_DT = {"f32": torch.float32, "f64": torch.float64, "f16": torch.float16}


def _rows(rs, n, c, kind):
    """Value patterns that stress different parts of the filter: well-separated mixture, one blob (tiny margins),
    quantised values (exact ties), sparse rows with exact zeros and duplicates, wide dynamic range."""
    if kind == "mixture":
        centers = rs.rand(min(32, max(2, n // 4 + 1)), c)
        x = centers[rs.randint(0, len(centers), n)] + 0.05 * rs.randn(n, c)
        x = np.maximum(x, 0.0)
    elif kind == "blob":
        x = 0.5 + 0.01 * rs.randn(n, c)
    elif kind == "quantised":
        x = rs.randint(0, 4, size=(n, c)).astype(np.float64) / 4.0
    elif kind == "sparse":
        x = rs.rand(n, c) * (rs.rand(n, c) < 0.3)
        if n > 4:
            x[rs.randint(0, n, n // 4)] = x[rs.randint(0, n, n // 4)]      # duplicated rows
    elif kind == "wild":   # a mixture with rows no arithmetic shortcut survives: NaN, infinities, huge and tiny values
        x = _rows(rs, n, c, "mixture")
        for value in (np.nan, np.inf, -np.inf, 1e30, -1e30, 1e-30, 6e4, 7e4):
            hit = rs.randint(0, n, max(1, n // 100))
            x[hit, rs.randint(0, c, len(hit))] = value
    else:   # "range"
        x = rs.rand(n, c) * np.exp(rs.uniform(-8, 3, size=(1, c)))
    return x


def _case(rs, max_work=6e7, wild=False):
    """(x [n, c] torch on the GPU -- possibly a strided view --, its float64 host twin, xdim, ydim, kind, dtype)."""
    xdim, ydim = int(rs.randint(1, 25)), int(rs.randint(1, 25))
    if rs.rand() < 0.35:
        xdim, ydim = 10, 10                                        # the register-resident shapes get their share
    k = xdim * ydim
    c = int(rs.choice([1, 2, 3, 5, 8, 13, 16, 22, 24, 31, 32, 33, 40, 64, 100, 128]))
    if rs.rand() < 0.3:
        c = int(rs.randint(1, 129))
    if rs.rand() < 0.06:
        c = int(rs.choice([48, 72, 104, 112]))                      # packed-K shapes (binary16 rows, K > 128)
    if rs.rand() < 0.04:
        c = int(rs.randint(129, 420))                               # wide rows: no filter, every row exact
    n_max = int(max(1, min(120000, max_work / (k * c))))
    n = int(rs.choice([1, 2, 63, 64, 65, 127, 129, 1000])) if rs.rand() < 0.25 else int(rs.randint(1, n_max + 1))
    n = min(n, n_max)
    dtype = str(rs.choice(["f32", "f32", "f64", "f16"]))
    dtype = os.environ.get("PXSOM_FUZZ_DTYPE", dtype)              # (pin the storage type of a sweep)
    kind = str(rs.choice(["mixture", "blob", "quantised", "sparse", "range"] + (["wild"] if wild else [])))
    host = _rows(rs, n, c, kind)
    pad = int(rs.choice([0, 0, 1, 2, 7]))
    buf = torch.zeros((n, c + pad), dtype=TORCH_DT[dtype])
    buf[:, :c] = torch.from_numpy(host).to(TORCH_DT[dtype])
    x = buf.cuda()[:, :c]                                          # row stride c + pad
    host = x.cpu().to(torch.float64).numpy()                       # the values the kernels really see
    return x, np.ascontiguousarray(host), xdim, ydim, kind, dtype


def _codebook(rs, host, k, kind):
    n, c = host.shape
    finite = host[np.isfinite(host).all(axis=1)] if host.size else host
    if len(finite) >= k and rs.rand() < 0.6:
        w = finite[rs.choice(len(finite), k, replace=False)].copy()   # rows as nodes: exact zero distances, duplicates
    else:
        w = rs.rand(k, c) * (np.abs(finite).max() if finite.size else 1.0)
    if k > 3 and rs.rand() < 0.3:
        w[rs.randint(0, k)] = w[rs.randint(0, k)]                  # a duplicated node: the first index must win
    return np.ascontiguousarray(w)


def _assert_sums_close(oracle, got, want, host, labels, k, tag, per_row=0.0):
    """Per-cluster sums to rounding, measured against sum |x| of the cluster's rows (values of both signs may cancel:
    1e30 - 1e30 + 7e4 depends on the order); entries whose rows are not all finite must agree in being non-finite.
    ``per_row``: absolute error allowed per row of the cluster on top (the fixed-point tables of the one-pass kernel)."""
    mag, cnt = oracle.cluster_sums(np.abs(host), labels, k)
    finite = np.isfinite(mag)
    assert np.array_equal(np.isfinite(got) & finite, np.isfinite(want) & finite), tag + ": finiteness of the sums"
    ok = finite & np.isfinite(want)
    err = np.abs(got[ok] - want[ok])
    bound = 1e-12 * mag[ok] + per_row * np.broadcast_to(cnt[:, None].astype(np.float64), mag.shape)[ok]
    assert (err <= bound).all(), tag + ": sums off by up to %.3g of the bound" % (err / np.maximum(bound, 1e-300)).max()


def test_fuzz_assign_and_sums(oracle):
    from ark_analysis_amd import som_device
    rs = np.random.RandomState(SEED)
    for case in range(CASES):
        x, host, xdim, ydim, kind, dtype = _case(rs, wild=True)
        k = xdim * ydim
        w = _codebook(rs, host, k, kind)
        tag = "case %d: n=%d c=%d k=%d %s %s ldx=%d" % (case, host.shape[0], host.shape[1], k, dtype, kind, x.stride(0))
        labels, _ = som_device.assign(x, torch.from_numpy(w).cuda())
        want, _ = oracle.map_data_to_nodes(w, host)
        got = labels.cpu().numpy()
        assert np.array_equal(got, want), tag + ": %d labels differ" % int((got != want).sum())
        sums, counts = som_device.cluster_sums(x, labels, k)
        ws, wc = oracle.cluster_sums(host, want, k)
        assert np.array_equal(counts.cpu().numpy(), wc), tag
        _assert_sums_close(oracle, sums.cpu().numpy(), ws, host, want, k, tag)
        lab2, s2, c2 = som_device.assign_sums(x, torch.from_numpy(w).cuda())
        assert np.array_equal(lab2.cpu().numpy(), want), tag + " (one pass)"
        assert np.array_equal(c2.cpu().numpy(), wc), tag + " (one pass)"
        # the register-resident shapes accumulate in fixed point: per value at most 2^-39 x rows-per-workgroup of the
        # codebook's largest magnitude; 2^-28 covers workgroups of up to 2 K rows
        wmax = float(np.abs(w[np.isfinite(w)]).max()) if np.isfinite(w).any() else 0.0
        _assert_sums_close(oracle, s2.cpu().numpy(), ws, host, want, k, tag + " (one pass)", per_row=wmax * 2.0 ** -28)
        if host.shape[0] >= 64 and rs.rand() < 0.3:
            # the same rows with neighbours that share their label (what images look like): runs of equal labels, shuffled --
            # the one-pass kernels sum such tiles along the row axis before they touch their tables
            order = np.argsort(want, kind="stable")
            run = int(rs.choice([3, 8, 16, 40, 1000]))
            pieces = max(1, host.shape[0] // run)
            perm = rs.permutation(pieces)
            idx = np.concatenate([(perm[:, None] * run + np.arange(run)[None, :]).reshape(-1), np.arange(pieces * run, host.shape[0])])
            idx = order[idx[idx < host.shape[0]]]
            xs = x[torch.from_numpy(idx).cuda()].contiguous()
            lab3, s3, c3 = som_device.assign_sums(xs, torch.from_numpy(w).cuda())
            assert np.array_equal(lab3.cpu().numpy(), want[idx]), tag + " (one pass, runs of %d)" % run
            assert np.array_equal(c3.cpu().numpy(), wc), tag + " (one pass, runs of %d)" % run
            _assert_sums_close(oracle, s3.cpu().numpy(), ws, host, want, k, tag + " (one pass, runs of %d)" % run, per_row=wmax * 2.0 ** -28)


def test_fuzz_batch_training(oracle):
    """Every mini-batch step checked on its own, from the state the GPU run itself holds: statistics of step g ==
    the oracle's for the codebook W_g the GPU searched with (labels bit-equal => counts equal, sums to rounding), and
    W_{g+1} == orc_batch_update(W_g, those statistics).  (Comparing only the final codebook with an independent oracle
    run is ill-posed on these value patterns: the two codebooks differ in their last bits -- different summation order
    of the row sums -- and one row that sits between two crowded nodes then changes sides, which moves the result by a
    whole row.  The named tests do make that end-to-end comparison, on data where it holds.)"""
    from ark_analysis_amd import som_device
    from ark_analysis_amd.flowsom import default_radius_range
    rs = np.random.RandomState(SEED + 1)
    for case in range(CASES):
        x, host, xdim, ydim, kind, dtype = _case(rs, max_work=1.5e7)
        n, c = host.shape
        k = xdim * ydim
        m = int(rs.choice([1, 2, 4, 8, 16]))
        passes = int(rs.choice([1, 1, 2]))
        w0 = _codebook(rs, host, k, kind)
        if n < 2:
            continue
        # the schedule: m equal steps, or m steps of unequal widths over a random number of phases
        from ark_analysis_amd.schedule import BatchSchedule
        if rs.rand() < 0.5:
            sch = BatchSchedule.equal(m)
        else:
            phases = int(rs.randint(m, 4 * m + 8))
            cuts = np.sort(rs.choice(np.arange(1, phases), size=m - 1, replace=False)) if m > 1 else np.empty(0, dtype=np.int64)
            sch = BatchSchedule(phases, [0] + [int(v) for v in cuts] + [phases])
        # binary64 rows: half of the cases train with the reproducible-statistics quantum (sums then equal the oracle's
        # on the quantised rows bit for bit)
        quantum = 0.0
        if dtype == "f64" and rs.rand() < 0.5 and np.isfinite(host).all():
            widest = int(np.diff(sch.edges).max())
            quantum = som_device.exact_sum_quantum(float(np.abs(host).max()), (n // sch.phases + 1) * max(widest, 1))
        total = m * passes
        alpha, radius = (0.05, 0.01), default_radius_range(xdim, ydim)
        for unfused in (False, True):
            tag = "case %d: n=%d c=%d grid=%dx%d %s %s steps=%d x %d phases=%d q=%g%s" % (
                case, n, c, xdim, ydim, dtype, kind, m, passes, sch.phases, quantum, " (unfused)" if unfused else "")
            st = som_device.BatchTrainState(n, c, xdim, ydim, sch, x.device)
            st.quantum = quantum
            st.wbuf[0].copy_(torch.from_numpy(w0))
            prev = None          # (W_g, statistics of step g, threshold, rate) of the step before
            for g in range(total + 1):
                if g < total:
                    som_device.batch_train_steps(x, st, g, g + 1, total, alpha, radius, unfused=unfused)
                    w_g = st.wbuf[g % 2].cpu().numpy().reshape(k, c)
                else:
                    out = torch.empty((k, c), dtype=torch.float64, device=x.device)
                    som_device.batch_train_finish(st, total, total, alpha, radius, out)
                    w_g = out.cpu().numpy()
                if prev is not None:
                    want_w = oracle.batch_update(prev[0], xdim, ydim, prev[1], prev[2], prev[3], prev[4])
                    # the update from the statistics the device holds is the oracle's bit for bit: window sums in the
                    # oracle's order, gain by the same chain of plain products (batch_gain; until round 5 it went through
                    # expm1 and was compared to 1e-13 of the channel's scale), no contraction
                    assert np.array_equal(w_g, want_w, equal_nan=True), tag + " update %d: %.3g" % (
                        g - 1, np.nanmax(np.abs(w_g - want_w)))
                if g == total:
                    break
                rows = host[sch.rows_of_step(n, g)].reshape(-1, c)
                want_l, _ = oracle.map_data_to_nodes(w_g, rows) if len(rows) else (np.empty(0, np.int32), None)
                want_s, want_c = oracle.cluster_sums(oracle.quantize(rows, quantum), want_l, k)
                ring = st.ring[g % 3].cpu().numpy()
                got_s, got_c = ring[:k * c].reshape(k, c), ring[k * c:]
                assert np.array_equal(got_c, want_c.astype(np.float64)), tag + " counts of step %d" % g
                if quantum:
                    assert np.array_equal(got_s, want_s), tag + " exact sums of step %d" % g
                else:
                    np.testing.assert_allclose(got_s, want_s, rtol=1e-12, atol=1e-300, err_msg=tag + " sums of step %d" % g)
                pos, span = sch.position(g), passes * sch.phases
                thr = radius[0] - (radius[0] - radius[1]) * pos / span
                a = alpha[0] - (alpha[0] - alpha[1]) * pos / span
                prev = (w_g, got_s.copy(), got_c.astype(np.int64), 0.5 if thr < 1.0 else thr, a)


def test_fuzz_batch_whole_runs_on_crowded_tables(oracle):
    """WHOLE runs against an independent oracle run (orc_som_batch_sched), bit for bit, on the tables where a last-bit
    difference would show: few rows (n <= 5 000), nodes drawn from the rows with duplicates among them (crowded first
    codebooks: a window of the first steps holds most of the table and its gain sits within an ulp of 1), values that are
    whole multiples of 1/256 (every partial sum is exact, so the order the rows are added in cannot matter -- what is left
    is the rule's own arithmetic: window sums, gain, update).  Until round 5 the gain went through expm1 and such runs
    parted ways with the oracle at their second step (scripts/debug/tail_case_probe.py)."""
    from ark_analysis_amd import som_device
    from ark_analysis_amd.flowsom import default_radius_range
    from ark_analysis_amd.schedule import BatchSchedule
    rs = np.random.RandomState(SEED + 7)
    for case in range(CASES):
        if rs.rand() < 0.5:
            xdim, ydim = 10, 10
            c = int(rs.choice([2, 4, 8, 16, 22, 32]))                  # the one-launch 10 x 10 step
        else:
            xdim, ydim = int(rs.randint(2, 15)), int(rs.randint(2, 15))
            c = int(rs.choice([1, 3, 8, 22, 33, 40, 100]))
        k = xdim * ydim
        n = int(rs.randint(max(k, 200), 5001))
        dtype = str(rs.choice(["f32", "f32", "f64", "f16"]))
        levels = int(rs.choice([4, 256, 2047]))                        # few levels: many equal rows and exact ties
        host = rs.randint(0, levels + 1, size=(n, c)).astype(np.float64) / 256.0
        host *= rs.rand(n, c) < rs.uniform(0.3, 1.0)                   # exact zeros, as in pixel tables
        pad = int(rs.choice([0, 0, 3]))
        buf = torch.zeros((n, c + pad), dtype=TORCH_DT[dtype])
        buf[:, :c] = torch.from_numpy(host).to(TORCH_DT[dtype])
        x = buf.cuda()[:, :c]
        assert np.array_equal(x.cpu().to(torch.float64).numpy(), host)  # every value is exact in the storage type
        w0 = host[rs.choice(n, k, replace=True)].copy()                # rows as nodes, with repeats: duplicate nodes
        w0[rs.randint(0, k)] = w0[rs.randint(0, k)]
        m = int(rs.choice([2, 4, 8, 22]))
        passes = int(rs.choice([1, 1, 2]))
        if m == 22:
            sch = BatchSchedule.two_phase(tail_phases_per_step=int(rs.choice([1, 2])))   # 120 / 240 phases, 22 steps
        elif rs.rand() < 0.5:
            sch = BatchSchedule.equal(m)
        else:
            phases = int(rs.randint(m, 4 * m + 8))
            cuts = np.sort(rs.choice(np.arange(1, phases), size=m - 1, replace=False))
            sch = BatchSchedule(phases, [0] + [int(v) for v in cuts] + [phases])
        alpha, radius = (0.05, 0.01), default_radius_range(xdim, ydim)
        want = oracle.som_batch_sched(host, w0, xdim, ydim, passes, alpha, radius, sch.phases, sch.edges)
        total = sch.steps * passes
        for unfused in (False, True):
            tag = "case %d: n=%d c=%d grid=%dx%d %s levels=%d steps=%d x %d phases=%d%s" % (
                case, n, c, xdim, ydim, dtype, levels, sch.steps, passes, sch.phases, " (unfused)" if unfused else "")
            st = som_device.BatchTrainState(n, c, xdim, ydim, sch, x.device, dtype=x.dtype)
            st.wbuf[0].copy_(torch.from_numpy(w0))
            som_device.batch_train_steps(x, st, 0, total, total, alpha, radius, unfused=unfused)
            out = torch.empty((k, c), dtype=torch.float64, device=x.device)
            som_device.batch_train_finish(st, total, total, alpha, radius, out)
            got = out.cpu().numpy()
            assert np.array_equal(got, want), tag + ": %d values differ, largest %.3g" % (
                int((got != want).sum()), float(np.abs(got - want).max()))


def test_fuzz_online_training(oracle):
    from ark_analysis_amd import som_device
    from ark_analysis_amd.flowsom import default_radius_range
    rs = np.random.RandomState(SEED + 2)
    for case in range(CASES):
        x, host, xdim, ydim, kind, dtype = _case(rs, max_work=4e6)
        n, c = host.shape
        k = xdim * ydim
        if k > 1024 or c > 128:
            continue
        rlen = int(rs.choice([1, 1, 2]))
        order = rs.randint(0, n, size=n * rlen).astype(np.int64)
        w0 = _codebook(rs, host, k, kind)
        alpha, radius = (0.05, 0.01), default_radius_range(xdim, ydim)
        tag = "case %d: n=%d c=%d grid=%dx%d %s %s rlen=%d" % (case, n, c, xdim, ydim, dtype, kind, rlen)
        want = oracle.som_online(host, w0, xdim, ydim, rlen, alpha, radius, order)
        w = torch.from_numpy(w0.copy()).cuda()
        som_device.train_online(x, w, xdim, ydim, rlen, alpha, radius, torch.from_numpy(order).cuda())
        assert np.array_equal(w.cpu().numpy(), want), tag


def test_fuzz_preprocessing(oracle):
    """K2-K5 and the small label kernels on random shapes: blur (binary64 and float32 semantics), row-sum filter +
    normalisation, column normalisation, non-zero / positive quantiles against numpy, pair histogram, relabel."""
    from scipy import ndimage
    from ark_analysis_amd import som_device as sd
    rs = np.random.RandomState(SEED + 3)
    dev = torch.device("cuda")
    for case in range(CASES):
        h, w, c = int(rs.randint(1, 90)), int(rs.randint(1, 90)), int(rs.randint(1, 9))
        sigma = float(rs.choice([0.5, 1.0, 2.0, 3.0]))
        f32 = bool(rs.rand() < 0.5)
        img = rs.gamma(0.5, 2.0, size=(h, w, c))
        img[rs.rand(h, w, c) < 0.4] = 0.0
        if f32:
            img = img.astype(np.float32).astype(np.float64)
        tag = "case %d: %dx%dx%d sigma %g f32=%s" % (case, h, w, c, sigma, f32)
        t = torch.from_numpy(img.copy()).to(dev)
        sd.gaussian_blur_hwc(t, sigma, f32_semantics=f32)
        want = oracle.gaussian_blur_hwc(img, sigma, f32=f32)
        assert np.array_equal(t.cpu().numpy(), want), tag + " blur"
        for j in range(c):     # ... and the oracle against scipy itself (images shorter than the kernel included)
            if f32:
                ref = ndimage.gaussian_filter(img[:, :, j].astype(np.float32), sigma)
                assert np.array_equal(want[:, :, j].astype(np.float32), ref), tag + " blur vs scipy (float32)"
            else:
                np.testing.assert_allclose(want[:, :, j], ndimage.gaussian_filter(img[:, :, j], sigma), rtol=1e-13,
                                           atol=1e-300, err_msg=tag + " blur vs scipy")
        # row-sum filter + row normalisation on the blurred pixels
        flat = np.ascontiguousarray(want.reshape(-1, c))
        thresh = float(rs.choice([0.0, np.median(flat.sum(1)), 1e30]))
        rows, kept = sd.rowsum_filter_normalize(torch.from_numpy(flat).to(dev), thresh, f32_semantics=f32)
        wr, wk = oracle.rowsum_filter_normalize(flat, thresh, sum_mode=2 if f32 else 0)
        assert np.array_equal(kept.cpu().numpy(), wk), tag + " kept rows"
        assert np.array_equal(rows.cpu().numpy(), wr), tag + " normalised rows"
        # quantiles of the kept rows' channels and the column normalisation by them
        if len(wk) > 0:
            q = float(rs.choice([0.05, 0.5, 0.99, 0.999]))
            mode = int(rs.choice([0, 1]))
            got_q = sd.quantile_nonzero(torch.from_numpy(wr).to(dev), q, keep_mode=mode).cpu().numpy()
            for j in range(c):
                col = wr[:, j]
                sel = col[col != 0] if mode == 0 else col[col > 0]
                if len(sel) == 0:
                    assert np.isnan(got_q[j]), tag + " empty quantile"
                else:
                    assert got_q[j] == np.quantile(sel, q), tag + " quantile of column %d" % j
            norm = np.where(np.isnan(got_q) | (got_q == 0), 1.0, got_q)
            out = sd.normalize_columns(torch.from_numpy(wr).to(dev), torch.from_numpy(norm).to(dev))
            assert np.array_equal(out.cpu().numpy(), wr / norm[None, :]), tag + " column normalisation"
        # label kernels
        n = int(rs.randint(1, 5000))
        na, nb = int(rs.randint(1, 300)), int(rs.randint(1, 40))
        a = rs.randint(-2, na + 2, size=n).astype(np.int32)
        b = rs.randint(-2, nb + 2, size=n).astype(np.int32)
        hist = sd.pair_histogram(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev), na, nb).cpu().numpy()
        assert np.array_equal(hist, oracle.pair_histogram(a, b, na, nb)), tag + " pair histogram"
        lut = rs.randint(0, 50, size=nb).astype(np.int32)
        got_l = sd.relabel(torch.from_numpy(b).to(dev), torch.from_numpy(lut).to(dev), fill=-7).cpu().numpy()
        inside = (b >= 0) & (b < nb)
        assert np.array_equal(got_l, np.where(inside, lut[np.clip(b, 0, nb - 1)], -7)), tag + " relabel"
        # float32 TIFF-side statistics: numpy's binary32 quantile and row sum of img / norm
        planes = img.astype(np.float32)
        q = float(rs.choice([0.05, 0.5, 0.99, 0.999]))
        got_q32 = sd.quantile_f32(torch.from_numpy(np.ascontiguousarray(planes.reshape(-1, c))).to(dev), q, keep_mode=1).cpu().numpy()
        for j in range(c):
            pos = planes[:, :, j][planes[:, :, j] > 0]
            if pos.size == 0:
                assert np.isnan(got_q32[j]), tag + " empty float32 quantile"
            else:
                assert got_q32[j] == np.quantile(pos, q), tag + " float32 quantile of channel %d" % j
        norm32 = rs.uniform(0.5, 3.0, size=c).astype(np.float32)
        got_rs = sd.scaled_rowsum_f32(torch.from_numpy(np.ascontiguousarray(planes.reshape(-1, c))).to(dev),
                                      torch.from_numpy(norm32).to(dev)).cpu().numpy()
        assert np.array_equal(got_rs.reshape(h, w), np.sum(planes / norm32.reshape([1, 1, c]), axis=-1)), tag + " scaled row sum"
        # cluster-id mask: relabel + scatter, the last row wins for a pixel listed twice
        m = int(rs.randint(1, h * w + 1))
        pos = rs.randint(0, h * w, size=m)
        lab = rs.randint(0, nb, size=m)
        lut16 = rs.randint(-30000, 30000, size=nb).astype(np.int32)
        mask, status = sd.cluster_mask(torch.from_numpy((pos // w).astype(np.int64)).to(dev),
                                       torch.from_numpy((pos % w).astype(np.int64)).to(dev),
                                       torch.from_numpy(lab.astype(np.int64)).to(dev), torch.from_numpy(lut16).to(dev), h, w)
        want_mask = np.zeros(h * w, dtype=np.int16)
        want_mask[pos] = lut16[lab].astype(np.int16)
        assert status == 0 and np.array_equal(mask.cpu().numpy().ravel(), want_mask), tag + " cluster mask"


def test_fuzz_arrow_labelling_path(tmp_path):
    """cluster_pixels' pandas-free labelling (arrow_assign.label_table, recycled host blocks) against the DataFrame
    route (PixelSOMCluster.assign_som_clusters) on random tables: chunked columns, extra columns, re-labelling."""
    import pandas as pd
    from ark_analysis_amd import arrow_assign
    from ark_analysis_amd.fov_tables import read_table, write_dataframe
    from ark_analysis_amd.phenotyping import cluster_helpers
    rs = np.random.RandomState(SEED + 4)
    blocks = arrow_assign.HostBlocks()
    for case in range(CASES):
        n, c = int(rs.randint(1, 6000)), int(rs.randint(1, 9))
        xdim, ydim = int(rs.randint(1, 7)), int(rs.randint(1, 7))
        chans = ["ch%d" % j for j in range(c)]
        df = pd.DataFrame(rs.gamma(0.5, 1.0, size=(n, c)), columns=chans)
        df["fov"] = "fov0"
        df["row_index"] = rs.randint(0, 100, n)
        df["column_index"] = rs.randint(0, 100, n)
        if rs.rand() < 0.5:
            df["label"] = rs.randint(0, 50, n)
        relabel = bool(rs.rand() < 0.3)
        if relabel:
            df["pixel_som_cluster"] = rs.randint(1, 5, n).astype(np.int32)
        norm = pd.DataFrame(rs.uniform(0.5, 2.0, size=(1, c)), columns=chans)
        weights = pd.DataFrame(rs.gamma(0.5, 1.0, size=(xdim * ydim, c)), columns=chans)
        for name, frame in (("norm", norm), ("weights", weights)):
            write_dataframe(frame, str(tmp_path / (name + ".feather")))
        (tmp_path / "sub").mkdir(exist_ok=True)
        write_dataframe(df.iloc[:1], str(tmp_path / "sub" / "fov0.feather"))
        som = cluster_helpers.PixelSOMCluster(str(tmp_path / "sub"), str(tmp_path / "norm.feather"),
                                              str(tmp_path / "weights.feather"), ["fov0"], chans, xdim=xdim, ydim=ydim)
        path = str(tmp_path / "table.feather")
        write_dataframe(df, path)
        table = read_table(path)                                     # 64 Ki-row record batches: chunked columns
        assert arrow_assign.applicable(som, table, not relabel)
        got, release, totals = arrow_assign.label_table(som, table, normalize=not relabel, blocks=blocks)
        got = got.to_pandas()
        if release is not None:
            release()
        # the per-cluster totals handed to generate_som_avg_files' cache == a groupby over the table as written
        feats, sums, counts = totals
        assert list(feats) == chans
        by = got.groupby("pixel_som_cluster")
        present = np.flatnonzero(counts) + 1
        assert np.array_equal(present, np.array(sorted(by.groups)))
        assert np.array_equal(counts[present - 1], by.size().to_numpy())
        np.testing.assert_allclose(sums[present - 1], by[chans].sum().to_numpy(), rtol=1e-12, atol=1e-300)
        seen_fast = set(som.som_clusters_seen)
        som.som_clusters_seen = set()
        base = df.drop(columns="pixel_som_cluster") if relabel else df
        want = som.assign_som_clusters(base, normalize_data=not relabel)
        pd.testing.assert_frame_equal(got, want)
        assert seen_fast == som.som_clusters_seen
    blocks.close()


def test_fuzz_front_ends(oracle):
    """The pyFlowSOM-shaped entry points on host arrays as callers hand them over: C / Fortran order, sliced views,
    float32 / float64 / integer dtypes, a single row; labels, *distances* and the seeded ``som`` against the oracle."""
    from ark_analysis_amd import flowsom
    rs = np.random.RandomState(SEED + 5)
    for case in range(CASES):
        xdim, ydim = int(rs.randint(1, 13)), int(rs.randint(1, 13))
        k = xdim * ydim
        c = int(rs.randint(1, 41))
        n = int(rs.randint(1, 3000))
        base = rs.gamma(0.5, 1.0, size=(n, c + 2))
        form = str(rs.choice(["c", "fortran", "sliced", "float32", "int"]))
        if form == "fortran":
            data = np.asfortranarray(base[:, :c])
        elif form == "sliced":
            data = base[:, 1:c + 1]
        elif form == "float32":
            data = base[:, :c].astype(np.float32)
        elif form == "int":
            data = (base[:, :c] * 10).astype(np.int64)
        else:
            data = np.ascontiguousarray(base[:, :c])
        host = np.ascontiguousarray(data, dtype=np.float64)
        w = rs.gamma(0.5, 1.0, size=(k, c))
        tag = "case %d: n=%d c=%d k=%d %s" % (case, n, c, k, form)
        labels, dists = flowsom.map_data_to_nodes(w, data)
        want_l, want_d = oracle.map_data_to_nodes(w, host)
        assert np.array_equal(labels, want_l), tag
        assert np.array_equal(dists, want_d), tag + " distances"       # sqrt of the binary64 sum, bit for bit
        one_l, one_d = flowsom.map_data_to_nodes(w, data[0])            # a single row, 1-D
        assert one_l[0] == want_l[0] and one_d[0] == want_d[0], tag + " single row"
        if n >= k and form != "int":
            seed = int(rs.randint(0, 1000))
            rlen = int(rs.choice([1, 2]))
            got = flowsom.som(data, xdim, ydim, rlen, (0.05, 0.01), seed=seed)
            init_idx, order = flowsom.som_init_and_order(n, k, rlen, seed)
            want = oracle.som_online(host, host[init_idx], xdim, ydim, rlen, (0.05, 0.01),
                                     flowsom.default_radius_range(xdim, ydim), order)
            assert np.array_equal(got, want), tag + " som(seed=%d, rlen=%d)" % (seed, rlen)
