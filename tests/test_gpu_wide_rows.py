"""Feature tables wider than the MFMA filter's 128 channels -- the reference's cell SOM takes any column count, e.g. the
400 pixel-SOM-cluster counts of a 20 x 20 pixel SOM (/root/reference/src/ark/phenotyping/cell_cluster_utils.py:63-192 ->
cluster_helpers.py:304-416).  Kernel level (labels, distances, per-cluster tables, online and batch training against
the oracle) and the drop-in train_cell_som / cluster_cells on a 400-column table.  `-m gpu` only."""
import os

import numpy as np
import pandas as pd
import pytest
import torch

from ark_analysis_amd import som_device as sd
from ark_analysis_amd.flowsom import default_radius_range
from ark_analysis_amd.schedule import BatchSchedule

pytestmark = pytest.mark.gpu


def _counts_table(n, c, seed, dtype=np.float64):
    """cell x cluster counts normalised by cell size: sparse, many exact ties (what create_c2pc_data produces)"""
    rs = np.random.RandomState(seed)
    x = rs.poisson(0.4, size=(n, c)).astype(np.float64) / rs.randint(50, 500, size=(n, 1))
    return np.ascontiguousarray(x.astype(dtype))


@pytest.mark.parametrize("n,c,k,dtype", [(20_001, 400, 100, np.float64), (5_000, 129, 100, np.float32), (3_000, 1024, 400, np.float32),
                                         (257, 400, 25, np.float16), (3, 200, 100, np.float64)])
def test_wide_assign_and_tables_match_oracle(gpu, oracle, n, c, k, dtype):
    x = _counts_table(max(n, k + 5), c, seed=c + n, dtype=dtype)[:n]
    x64 = x.astype(np.float64)
    rs = np.random.RandomState(1)
    w = _counts_table(k + 5, c, seed=7)[:k] + 1e-4 * rs.rand(k, c)
    w[k - 1] = w[0]                                   # a duplicate node: the first one wins
    if n > 10:
        x[5] = x[4]
        x[7, 3] = np.nan                              # NaN row -> label 0
        x64 = x.astype(np.float64)
    xd, wd = torch.from_numpy(x).to(gpu), torch.from_numpy(w).to(gpu)
    labels, dists = sd.assign(xd, wd, want_dists=True)
    want_l, want_d = oracle.map_data_to_nodes(w, x64)
    np.testing.assert_array_equal(labels.cpu().numpy(), want_l)
    ok = want_l > 0
    np.testing.assert_array_equal(dists.cpu().numpy()[ok], want_d[ok])
    assert sd.last_exact_rows(sd.assign.last_workspace) == n
    sums, counts = sd.cluster_sums(xd, labels, k)
    s, cnt = oracle.cluster_sums(x64, want_l, k)
    np.testing.assert_array_equal(counts.cpu().numpy(), cnt)
    np.testing.assert_allclose(sums.cpu().numpy(), s, rtol=1e-12, atol=1e-300)
    l1, s1, c1 = sd.assign_sums(xd, wd)               # one entry point, two kernels for these shapes
    assert torch.equal(l1, labels) and torch.equal(c1, counts)


def test_wide_online_and_batch_training_match_oracle(gpu, oracle):
    n, c, xdim, ydim = 6_000, 400, 10, 10
    k = xdim * ydim
    x = _counts_table(n, c, seed=3)
    rs = np.random.RandomState(2)
    w0 = np.ascontiguousarray(x[rs.choice(n, k, replace=False)])
    rr = default_radius_range(xdim, ydim)
    xd = torch.from_numpy(x).to(gpu)
    order = rs.randint(0, n, size=n).astype(np.int64)
    wd = torch.from_numpy(w0.copy()).to(gpu)
    sd.train_online(xd, wd, xdim, ydim, 1, (0.05, 0.01), rr, torch.from_numpy(order).to(gpu))
    assert np.array_equal(wd.cpu().numpy(), oracle.som_online(x, w0, xdim, ydim, 1, (0.05, 0.01), rr, order))
    # batch rule, step by step on the state the GPU run itself holds (count tables are full of near-ties: two independent
    # runs may part ways on the last bits of a sum, each step is still the oracle's step for the codebook it searched with)
    from ark_analysis_amd.distributed import batch_schedule
    sch = BatchSchedule(12, [0, 5, 8, 9, 12])
    st = sd.BatchTrainState(n, c, xdim, ydim, sch, gpu, dtype=xd.dtype)
    st.wbuf[0].copy_(torch.from_numpy(w0))
    w_prev = s_prev = cnt_prev = None
    for g in range(sch.steps):
        sd.batch_train_steps(xd, st, g, g + 1, sch.steps, (0.05, 0.01), rr)
        w_g = st.wbuf[g % 2].cpu().numpy()
        if g > 0:
            thr, alpha = batch_schedule(sch.position(g - 1), sch.phases, (0.05, 0.01), rr)
            np.testing.assert_allclose(w_g, oracle.batch_update(w_prev, xdim, ydim, s_prev, cnt_prev, thr, alpha), rtol=1e-12, atol=1e-300)
        rows = x[sch.rows_of_step(n, g)]
        lab, _ = oracle.map_data_to_nodes(w_g, rows)
        s, cnt = oracle.cluster_sums(rows, lab, k)
        ring = st.ring[g % 3].cpu().numpy()
        np.testing.assert_array_equal(ring[k * c:], cnt.astype(np.float64))
        np.testing.assert_allclose(ring[: k * c].reshape(k, c), s, rtol=1e-12, atol=1e-300)
        w_prev, s_prev, cnt_prev = w_g, ring[: k * c].reshape(k, c).copy(), cnt
    wb = torch.empty((k, c), dtype=torch.float64, device=gpu)
    sd.batch_train_finish(st, sch.steps, sch.steps, (0.05, 0.01), rr, wb)
    thr, alpha = batch_schedule(sch.position(sch.steps - 1), sch.phases, (0.05, 0.01), rr)
    np.testing.assert_allclose(wb.cpu().numpy(), oracle.batch_update(w_prev, xdim, ydim, s_prev, cnt_prev, thr, alpha), rtol=1e-12, atol=1e-300)


def test_cell_som_on_400_pixel_cluster_columns(gpu, oracle, tmp_path):
    """train_cell_som + cluster_cells on a cell table with the 400 count columns of a 20 x 20 pixel SOM (the reference's
    pixel_cluster_col='pixel_som_cluster' route): the codebook equals the oracle's online run for the same seed, the
    labels the oracle's for that codebook."""
    from ark_analysis_amd import flowsom
    from ark_analysis_amd.phenotyping import cell_som_clustering
    n, c = 1_500, 400
    cols = ["pixel_som_cluster_%d" % i for i in range(1, c + 1)]
    cell = pd.DataFrame(_counts_table(n, c, seed=11), columns=cols)
    cell["fov"] = np.where(np.arange(n) % 2 == 0, "fov0", "fov1")
    cell["segmentation_label"] = np.arange(n)
    cell["cell_size"] = np.random.RandomState(0).randint(50, 500, size=n)
    td = str(tmp_path)
    open(os.path.join(td, "cell_table.csv"), "w").write("x\n")
    cobj = cell_som_clustering.train_cell_som(["fov0", "fov1"], td, os.path.join(td, "cell_table.csv"), cols, cell.copy(), seed=42)
    res = cell_som_clustering.cluster_cells(td, cobj, cols)
    normed = cobj.cell_data[cols].values.astype(np.float64)
    init_idx, order = flowsom.som_init_and_order(n, 100, 1, 42)
    want_w = oracle.som_online(normed, normed[init_idx].copy(), 10, 10, 1, (0.05, 0.01), default_radius_range(10, 10), order)
    np.testing.assert_array_equal(cobj.weights.values, want_w)
    want_l, _ = oracle.map_data_to_nodes(want_w, normed)
    np.testing.assert_array_equal(res["cell_som_cluster"].values, want_l)
