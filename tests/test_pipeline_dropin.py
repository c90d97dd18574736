"""The drop-in pipeline functions (ark_analysis_amd.phenotyping.*) against fixtures produced by
the REFERENCE's own train_pixel_som / cluster_pixels / generate_som_avg_files / train_cell_som /
cluster_cells (tests/golden/g7_*.npz; SOM arithmetic inside = oracle of record), plus mirrors of
the reference's own unit tests for these functions.  Every test runs on CPU with the oracle
standing in for the three device entry points and, under -m gpu, on the real HIP path."""
import os
import warnings

import numpy as np
import pandas as pd
import pytest

from ark_analysis_amd.phenotyping import (cell_som_clustering, cluster_helpers, pixel_cluster_utils,
                                          pixel_som_clustering)
from ark_analysis_amd.phenotyping.cluster_helpers import read_dataframe, write_dataframe

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CHANS = ["chan%d" % i for i in range(4)]
FOVS = ["fov0", "fov1", "fov2"]


def _build_pixel_dirs(td, g):
    os.mkdir(os.path.join(td, "pixel_mat_data"))
    os.mkdir(os.path.join(td, "pixel_mat_subsetted"))
    write_dataframe(pd.DataFrame(g["norm"][None, :], columns=CHANS),
                    os.path.join(td, "post_rowsum_chan_norm.feather"))
    for fov in FOVS:
        df = pd.DataFrame(g["data_" + fov], columns=CHANS)
        df["fov"] = fov
        meta = g["meta_" + fov]
        df["row_index"], df["column_index"], df["label"] = meta[:, 0], meta[:, 1], meta[:, 2]
        write_dataframe(df, os.path.join(td, "pixel_mat_data", fov + ".feather"))
        write_dataframe(df.iloc[g["subidx_" + fov]], os.path.join(td, "pixel_mat_subsetted", fov + ".feather"))


def test_pixel_pipeline_matches_reference_run(som_backend, tmp_path, capsys):
    g = np.load(os.path.join(GOLD, "g7_pixel_pipeline.npz"))
    td = str(tmp_path)
    _build_pixel_dirs(td, g)
    obj = pixel_som_clustering.train_pixel_som(FOVS, CHANS, td, num_passes=1, seed=42)
    pixel_som_clustering.cluster_pixels(FOVS, td, obj)
    pixel_som_clustering.generate_som_avg_files(FOVS, CHANS, td, obj, data_dir="pixel_mat_data")
    assert capsys.readouterr().out == str(g["stdout"])
    # trained codebook: bit-equal (exact online mode), columns in training order
    assert list(obj.weights.columns) == CHANS
    np.testing.assert_array_equal(obj.weights.values, g["weights"])
    np.testing.assert_array_equal(read_dataframe(os.path.join(td, "pixel_som_weights.feather")).values,
                                  g["weights"])
    for fov in FOVS:
        res = read_dataframe(os.path.join(td, "pixel_mat_data", fov + ".feather"))
        np.testing.assert_array_equal(res["pixel_som_cluster"].values, g["labels_" + fov])
        np.testing.assert_array_equal(res[CHANS].values, g["normed_" + fov])   # 99.9 %-normalised write-back
    assert not os.path.exists(os.path.join(td, "pixel_mat_data_temp"))
    assert sorted(int(v) for v in obj.som_clusters_seen) == list(g["clusters_seen"])
    avg = pd.read_csv(os.path.join(td, "pixel_channel_avg_som_cluster.csv"))
    np.testing.assert_array_equal(avg["pixel_som_cluster"].values, g["avg_clusters"])
    np.testing.assert_array_equal(avg["count"].values, g["avg_count"])
    np.testing.assert_allclose(avg[CHANS].values, g["avg_means"], rtol=1e-12, atol=0)


def test_pixel_pipeline_batch_mode_matches_fixture(som_backend, tmp_path, capsys):
    """train_pixel_som(..., train_mode="batch", batch_steps=8) -> cluster_pixels -> generate_som_avg_files against
    the reference's own pipeline run with the build's batch rule underneath (g7b; oracle of record orc_som_batch)."""
    g = np.load(os.path.join(GOLD, "g7b_pixel_pipeline_batch.npz"))
    td = str(tmp_path)
    _build_pixel_dirs(td, g)
    obj = pixel_som_clustering.train_pixel_som(FOVS, CHANS, td, num_passes=1, seed=42, train_mode="batch",
                                               batch_steps=8)
    assert obj.train_mode == "batch" and obj.batch_steps == 8
    pixel_som_clustering.cluster_pixels(FOVS, td, obj)
    pixel_som_clustering.generate_som_avg_files(FOVS, CHANS, td, obj, data_dir="pixel_mat_data")
    assert capsys.readouterr().out == str(g["stdout"])
    assert list(obj.weights.columns) == CHANS
    # batch rule on binary64 tables: the statistics are exact sums of quantised rows (order-free) and the gain is a chain of
    # plain products on both sides (batch_gain, round 5): the oracle's codebook bit for bit
    np.testing.assert_array_equal(obj.weights.values, g["weights"])
    np.testing.assert_array_equal(read_dataframe(os.path.join(td, "pixel_som_weights.feather")).values,
                                  obj.weights.values)
    from tests import oracle_binding as ob
    for fov in FOVS:
        res = read_dataframe(os.path.join(td, "pixel_mat_data", fov + ".feather"))
        np.testing.assert_array_equal(res[CHANS].values, g["normed_" + fov])
        want, _ = ob.map_data_to_nodes(obj.weights.values, res[CHANS].values)
        np.testing.assert_array_equal(res["pixel_som_cluster"].values, want)
        np.testing.assert_array_equal(res["pixel_som_cluster"].values, g["labels_" + fov])
    # the reference pins same-seed retraining (tests/phenotyping/cluster_helpers_test.py:323-332): a second run on the
    # same binary64 tables gives the same bits -- weights and labels
    first_w = obj.weights.values.copy()
    first_l = {fov: read_dataframe(os.path.join(td, "pixel_mat_data", fov + ".feather"))["pixel_som_cluster"].values for fov in FOVS}
    td2 = str(tmp_path / "again")
    os.mkdir(td2)
    _build_pixel_dirs(td2, g)
    obj2 = pixel_som_clustering.train_pixel_som(FOVS, CHANS, td2, num_passes=1, seed=42, train_mode="batch", batch_steps=8)
    pixel_som_clustering.cluster_pixels(FOVS, td2, obj2)
    capsys.readouterr()
    np.testing.assert_array_equal(obj2.weights.values, first_w)
    for fov in FOVS:
        np.testing.assert_array_equal(read_dataframe(os.path.join(td2, "pixel_mat_data", fov + ".feather"))["pixel_som_cluster"].values,
                                      first_l[fov])
    if som_backend == "oracle":
        np.testing.assert_array_equal(obj.weights.values, g["weights"])
    with pytest.raises(ValueError, match="train_mode"):
        pixel_som_clustering.train_pixel_som(FOVS, CHANS, td, train_mode="minibatch")
    with pytest.raises(ValueError, match="batch_steps"):
        pixel_som_clustering.train_pixel_som(FOVS, CHANS, td, train_mode="batch", batch_steps=0)


def test_cell_pipeline_matches_reference_run(som_backend, tmp_path, capsys):
    g = np.load(os.path.join(GOLD, "g7_cell_pipeline.npz"))
    cols = ["pixel_meta_cluster_%d" % i for i in range(1, 9)]
    cell = pd.DataFrame(g["cell"], columns=cols)
    cell["fov"] = g["fov"]
    cell["segmentation_label"] = np.arange(len(cell))
    cell["cell_size"] = g["cell_size"]
    td = str(tmp_path)
    open(os.path.join(td, "cell_table.csv"), "w").write("x\n")
    cobj = cell_som_clustering.train_cell_som(["fov0", "fov1"], td, os.path.join(td, "cell_table.csv"),
                                              cols, cell.copy(), seed=42)
    res = cell_som_clustering.cluster_cells(td, cobj, cols)
    assert capsys.readouterr().out == str(g["stdout"])
    np.testing.assert_array_equal(cobj.weights.values, g["weights"])
    np.testing.assert_array_equal(res[cols].values, g["normed"])
    np.testing.assert_array_equal(res["cell_som_cluster"].values, g["labels"])
    # second call without overwrite returns immediately (cell_som_clustering.py:104-108)
    cell_som_clustering.cluster_cells(td, cobj, cols)
    assert capsys.readouterr().out == "SOM clusters already assigned to each cell\n"
    cell_som_clustering.generate_som_avg_files(td, res, cols, "avgs.csv")
    avg = pd.read_csv(os.path.join(td, "avgs.csv"))
    assert list(avg.columns) == ["cell_som_cluster"] + cols + ["count"]
    assert avg["count"].sum() == len(res)


# ---- mirrors of the reference's own tests (tests/phenotyping/cluster_helpers_test.py:286-420) ----
@pytest.fixture()
def pixel_objs(som_backend, tmp_path):
    rs = np.random.RandomState(0)
    cols = [f"Marker{i}" for i in range(1, 7)]
    fovs = [f"fov{i}" for i in range(3)]
    sub = tmp_path / "pixel_mat_subsetted"
    sub.mkdir()
    for fov in fovs:
        df = pd.DataFrame(rs.rand(100, 6), columns=cols)
        df["fov"] = fov
        df["row_index"] = rs.randint(0, 10, 100)
        df["column_index"] = rs.randint(0, 10, 100)
        df["label"] = rs.randint(0, 5, 100)
        write_dataframe(df, str(sub / f"{fov}.feather"))
    norm_path = str(tmp_path / "norm.feather")
    write_dataframe(pd.DataFrame(np.expand_dims(np.repeat(0.5, 6), 0), columns=cols), norm_path)
    weights_path = str(tmp_path / "weights_new.feather")
    obj = cluster_helpers.PixelSOMCluster(str(sub), norm_path, weights_path, fovs, cols, xdim=20, ydim=10)
    return obj, cols, rs


def test_reference_normalize_data(pixel_objs):
    obj, cols, rs = pixel_objs
    meta = ["fov", "row_index", "column_index", "label"]
    ext = pd.DataFrame(rs.rand(1000, 10), columns=cols + meta)
    out = obj.normalize_data(ext)
    assert np.allclose(ext[cols].values / 0.5, out[cols].values)


def test_reference_train_som_behaviour(pixel_objs):
    obj, cols, rs = pixel_objs
    obj.train_som()
    assert os.path.exists(obj.weights_path)
    assert list(obj.weights.columns.values) == cols          # order preserved
    assert obj.weights.shape == (200, 6)
    with pytest.warns(UserWarning, match="Pixel SOM already trained on specified markers"):
        obj.train_som()
    first = obj.weights.copy()
    with pytest.warns(UserWarning, match="Overwrite flag set, retraining SOM"):
        obj.train_som(overwrite=True)
    assert np.allclose(first.values, obj.weights.values)      # same seed -> same weights
    obj.columns = cols[:-1]
    with pytest.warns(UserWarning, match="New markers specified, retraining"):
        obj.train_som()
    assert list(obj.weights.columns.values) == cols[:-1]


def test_reference_assign_som_clusters(pixel_objs):
    obj, cols, rs = pixel_objs
    obj.train_som()
    meta = ["fov", "row_index", "column_index", "label"]
    for npp in (10, 10000):
        ext = pd.DataFrame(rs.rand(1000, 10), columns=cols[::-1] + meta)   # shuffled column order
        out = obj.assign_som_clusters(ext, num_parallel_pixels=npp)
        assert out["pixel_som_cluster"].dtype.kind in "iu"
        assert out["pixel_som_cluster"].min() >= 1 and out["pixel_som_cluster"].max() <= 200
        again = obj.assign_som_clusters(out.drop(columns="pixel_som_cluster"), normalize_data=False,
                                        num_parallel_pixels=npp)
        assert np.array_equal(again[cols].values, out[cols].values)
        assert np.array_equal(again["pixel_som_cluster"].values, out["pixel_som_cluster"].values)
    with pytest.raises(ValueError):
        obj.generate_som_clusters(ext, num_parallel_obs=0)
    empty = obj.generate_som_clusters(ext.iloc[:0])
    assert empty.shape == (0,)


# ---- mirrors of tests/phenotyping/pixel_som_clustering_test.py -------------------------------
def test_reference_train_pixel_som_errors(som_backend, tmp_path):
    g = np.load(os.path.join(GOLD, "g7_pixel_pipeline.npz"))
    td = str(tmp_path)
    _build_pixel_dirs(td, g)
    with pytest.raises(FileNotFoundError):
        pixel_som_clustering.train_pixel_som(FOVS, CHANS, td, subset_dir="bad_path")
    with pytest.raises(FileNotFoundError):
        pixel_som_clustering.train_pixel_som(FOVS, CHANS, td, norm_vals_name="bad.feather")
    with pytest.raises(ValueError):
        pixel_som_clustering.train_pixel_som(["fov0", "fov9"], CHANS, td)
    with pytest.raises(ValueError):
        pixel_som_clustering.train_pixel_som(FOVS, ["chan0", "nope"], td)


@pytest.mark.parametrize("multiprocess", [False, True])
def test_reference_cluster_pixels_behaviour(som_backend, tmp_path, capsys, multiprocess):
    g = np.load(os.path.join(GOLD, "g7_pixel_pipeline.npz"))
    td = str(tmp_path)
    _build_pixel_dirs(td, g)
    norm_path = os.path.join(td, "post_rowsum_chan_norm.feather")
    untrained = cluster_helpers.PixelSOMCluster(os.path.join(td, "pixel_mat_subsetted"), norm_path,
                                                os.path.join(td, "none.feather"), FOVS, CHANS)
    with pytest.raises(ValueError, match="Using untrained pixel_pysom object"):
        pixel_som_clustering.cluster_pixels(FOVS, td, untrained)
    obj = pixel_som_clustering.train_pixel_som(FOVS, CHANS, td)
    capsys.readouterr()
    # corrupt one FOV: reported, skipped, dropped from the output directory
    with open(os.path.join(td, "pixel_mat_data", "fov1.feather"), "w") as f:
        f.write("baddatabaddatabaddata")
    pixel_som_clustering.cluster_pixels(FOVS, td, obj, multiprocess=multiprocess, batch_size=2)
    out = capsys.readouterr().out
    assert "The data for FOV fov1 has been corrupted, skipping\n" in out
    assert not os.path.exists(os.path.join(td, "pixel_mat_data_temp"))
    assert sorted(os.listdir(os.path.join(td, "pixel_mat_data"))) == ["fov0.feather", "fov2.feather"]
    for fov in ("fov0", "fov2"):
        res = read_dataframe(os.path.join(td, "pixel_mat_data", fov + ".feather"))
        assert res["pixel_som_cluster"].max() <= 100
        np.testing.assert_array_equal(res["pixel_som_cluster"].values, g["labels_" + fov])
    # nothing left to do
    pixel_som_clustering.cluster_pixels(["fov0", "fov2"], td, obj, multiprocess=multiprocess)
    assert capsys.readouterr().out == "There are no more FOVs to assign SOM labels to, skipping\n"
    # overwrite: data on disk is already normalised -> labels unchanged
    pixel_som_clustering.cluster_pixels(["fov0", "fov2"], td, obj, multiprocess=multiprocess, overwrite=True)
    out = capsys.readouterr().out
    assert out.startswith("Overwrite flag set, reassigning SOM cluster labels to all FOVs\n")
    res = read_dataframe(os.path.join(td, "pixel_mat_data", "fov0.feather"))
    np.testing.assert_array_equal(res["pixel_som_cluster"].values, g["labels_fov0"])
    np.testing.assert_array_equal(res[CHANS].values, g["normed_fov0"])


def test_cluster_pixels_restarts_from_staged_tables(som_backend, tmp_path, capsys):
    """An interrupted run left <data_dir>_temp behind with one FOV done: only the others are processed,
    the staged table is kept as it is, and the directories are swapped at the end
    (reference behaviour: pixel_cluster_utils.py:419-478 + pixel_som_clustering.py:231-251)."""
    import shutil
    from ark_analysis_amd.fov_tables import FovTableDir
    g = np.load(os.path.join(GOLD, "g7_pixel_pipeline.npz"))
    td = str(tmp_path)
    _build_pixel_dirs(td, g)
    obj = pixel_som_clustering.train_pixel_som(FOVS, CHANS, td)
    # a complete run elsewhere provides the table an interrupted run would have staged
    shutil.copytree(os.path.join(td, "pixel_mat_data"), os.path.join(td, "full"))
    pixel_som_clustering.cluster_pixels(FOVS, td, obj, data_dir="full")
    os.mkdir(os.path.join(td, "pixel_mat_data_temp"))
    shutil.copy(os.path.join(td, "full", "fov1.feather"), os.path.join(td, "pixel_mat_data_temp", "fov1.feather"))
    assert sorted(FovTableDir(os.path.join(td, "pixel_mat_data")).pending("pixel_som_cluster")) == ["fov0", "fov2"]
    capsys.readouterr()
    pixel_som_clustering.cluster_pixels(FOVS, td, obj)
    out = capsys.readouterr().out
    assert "Restarting SOM label assignment from fov " in out and "2 fovs left to process\n" in out
    assert out.endswith("Processed 2 fovs\n")
    assert not os.path.exists(os.path.join(td, "pixel_mat_data_temp"))
    for fov in FOVS:
        res = read_dataframe(os.path.join(td, "pixel_mat_data", fov + ".feather"))
        np.testing.assert_array_equal(res["pixel_som_cluster"].values, g["labels_" + fov])
        np.testing.assert_array_equal(res[CHANS].values, g["normed_" + fov])


@pytest.mark.gpu
def test_arrow_fast_path_equals_dataframe_path(gpu, tmp_path):
    """cluster_pixels' pandas-free labelling writes exactly the table the DataFrame path writes
    (values, dtypes, column order, index), for first labelling and for re-labelling."""
    from ark_analysis_amd import arrow_assign, fov_tables
    g = np.load(os.path.join(GOLD, "g7_pixel_pipeline.npz"))
    td = str(tmp_path)
    _build_pixel_dirs(td, g)
    obj = pixel_som_clustering.train_pixel_som(FOVS, CHANS, td)
    for fov in FOVS:
        src = os.path.join(td, "pixel_mat_data", fov + ".feather")
        arrow_in = fov_tables.read_table(src)
        assert arrow_assign.applicable(obj, arrow_in, True)
        obj.som_clusters_seen = set()
        fast = arrow_assign.label_table(obj, arrow_in, normalize=True)
        seen_fast = set(obj.som_clusters_seen)
        obj.som_clusters_seen = set()
        slow = obj.assign_som_clusters(fov_tables.read_dataframe(src))
        assert seen_fast == {int(v) for v in obj.som_clusters_seen}
        fov_tables.write_dataframe(fast, os.path.join(td, "fast.feather"))
        fov_tables.write_dataframe(slow, os.path.join(td, "slow.feather"))
        a, b = read_dataframe(os.path.join(td, "fast.feather")), read_dataframe(os.path.join(td, "slow.feather"))
        pd.testing.assert_frame_equal(a, b, check_exact=True)
        np.testing.assert_array_equal(a["pixel_som_cluster"].values, g["labels_" + fov])
        np.testing.assert_array_equal(a[CHANS].values, g["normed_" + fov])
        # re-labelling an already labelled, already normalised table
        again = arrow_assign.label_table(obj, fov_tables.read_table(os.path.join(td, "fast.feather")), normalize=False)
        fov_tables.write_dataframe(again, os.path.join(td, "again.feather"))
        pd.testing.assert_frame_equal(read_dataframe(os.path.join(td, "again.feather")), a, check_exact=True)
    # tables the fast path does not cover fall back (float32 channel)
    odd = fov_tables.read_table(os.path.join(td, "pixel_mat_data", "fov0.feather"))
    import pyarrow as pa
    odd = odd.set_column(0, CHANS[0], odd.column(CHANS[0]).cast(pa.float32()))
    assert not arrow_assign.applicable(obj, odd, True)


@pytest.mark.gpu
def test_cluster_channel_avg_arrow_route_equals_dataframe_route(gpu, tmp_path, capsys):
    """compute_pixel_cluster_channel_avg straight from the Arrow tables == through DataFrames: integer and
    float-typed cluster ids, ids that are not 1..K, a float32 channel (falls back per table), a damaged file."""
    from ark_analysis_amd import fov_tables
    rs = np.random.RandomState(12)
    td = str(tmp_path)
    os.mkdir(os.path.join(td, "pixel_mat_data"))
    chans = ["chan%d" % i for i in range(5)]
    fovs = ["fov%d" % i for i in range(4)]
    for i, fov in enumerate(fovs):
        n = 3000 + 17 * i
        df = pd.DataFrame(rs.rand(n, 5), columns=chans)
        df["fov"] = fov
        df["pixel_som_cluster"] = rs.choice([2, 3, 5, 8, 13, 40], size=n)
        df["pixel_meta_cluster"] = rs.choice([1.0, 2.0, 7.0], size=n)          # float-typed ids
        if i == 2:
            df["chan1"] = df["chan1"].astype(np.float32)                        # not covered: per-table fallback
        fov_tables.write_dataframe(df, os.path.join(td, "pixel_mat_data", fov + ".feather"))
    with open(os.path.join(td, "pixel_mat_data", "fov3.feather"), "r+b") as f:  # damaged table
        f.truncate(1000)
    for col in ("pixel_som_cluster", "pixel_meta_cluster"):
        fast = pixel_cluster_utils.compute_pixel_cluster_channel_avg(fovs, chans, td, col, None, keep_count=True)
        out_fast = capsys.readouterr().out
        real = pixel_cluster_utils._DEVICE_SUMS
        pixel_cluster_utils._DEVICE_SUMS = None            # DataFrame route (same kernel underneath)
        try:
            slow = pixel_cluster_utils.compute_pixel_cluster_channel_avg(fovs, chans, td, col, None, keep_count=True)
        finally:
            pixel_cluster_utils._DEVICE_SUMS = real
        assert out_fast == capsys.readouterr().out == "The data for FOV fov3 has been corrupted, skipping\n"
        assert list(fast.columns) == list(slow.columns) and [str(d) for d in fast.dtypes] == [str(d) for d in slow.dtypes]
        np.testing.assert_array_equal(fast[col].values, slow[col].values)
        np.testing.assert_array_equal(fast["count"].values, slow["count"].values)
        np.testing.assert_allclose(fast[chans].values, slow[chans].values, rtol=1e-13, atol=0)


@pytest.mark.parametrize("cluster_col", ["pixel_som_cluster", "pixel_meta_cluster_rename"])
def test_create_c2pc_data_matches_reference_run(som_backend, tmp_path, cluster_col):
    """cell x pixel-cluster counts and their cell_size-normalised twin against the reference's own
    create_c2pc_data on the same tables (tests/golden/g8_c2pc.npz)."""
    from ark_analysis_amd.phenotyping import cell_cluster_utils
    g = np.load(os.path.join(GOLD, "g8_c2pc.npz"))
    fovs = ["fov0", "fov1", "fov2"]
    pix = tmp_path / "pixel_mat_data"
    pix.mkdir()
    for i, fov in enumerate(fovs):
        n = len(g["lab_" + fov])
        df = pd.DataFrame({"chan0": np.zeros(n)})
        df["fov"] = fov
        df["segmentation_label" if i == 1 else "label"] = g["lab_" + fov]
        df["pixel_som_cluster"] = g["som_" + fov]
        meta = g["meta_" + fov]
        df["pixel_meta_cluster_rename"] = meta.astype(np.float64) if i == 2 else meta
        write_dataframe(df, str(pix / (fov + ".feather")))
    cell = pd.DataFrame({"fov": g["cell_fov"], "label": g["cell_label"], "cell_size": g["cell_size"],
                         "extra": g["cell_extra"]})
    cell_path = str(tmp_path / "cell_table.csv")
    cell.to_csv(cell_path, index=False)
    with warnings.catch_warnings(record=True) as wl:
        warnings.simplefilter("always")
        counts, normed = cell_cluster_utils.create_c2pc_data(fovs, str(pix), cell_path, cluster_col)
    assert [str(w.message) for w in wl if "Pixel clusters" in str(w.message)] == list(g[cluster_col + "_warnings"])
    for tag, frame in (("counts", counts), ("normed", normed)):
        assert list(frame.columns) == list(g[f"{cluster_col}_{tag}_columns"])
        assert [str(t) for t in frame.dtypes] == list(g[f"{cluster_col}_{tag}_dtypes"])
        assert list(frame["fov"]) == list(g[f"{cluster_col}_{tag}_fov"])
        assert list(frame.index) == list(range(len(frame)))
        np.testing.assert_array_equal(frame.drop(columns="fov").values.astype(np.float64),
                                      g[f"{cluster_col}_{tag}_values"])
    with pytest.raises(ValueError):
        cell_cluster_utils.create_c2pc_data(fovs, str(pix), cell_path, "pixel_meta_cluster")


def test_create_c2pc_data_with_named_clusters(som_backend, tmp_path):
    """The default column, pixel_meta_cluster_rename, holds user-assigned names ('CD4_T', 'tumor 2', ...): used as
    they are, like the reference does (tests/golden/g8s_c2pc_named.npz = the reference's own output)."""
    from ark_analysis_amd.phenotyping import cell_cluster_utils
    g = np.load(os.path.join(GOLD, "g8s_c2pc_named.npz"))
    fovs = ["fov0", "fov1"]
    pix = tmp_path / "pixel_mat_data"
    pix.mkdir()
    for fov in fovs:
        n = len(g["lab_" + fov])
        df = pd.DataFrame({"chan0": np.zeros(n)})
        df["fov"] = fov
        df["label"] = g["lab_" + fov]
        df["pixel_som_cluster"] = 1
        df["pixel_meta_cluster_rename"] = g["names"][g["code_" + fov]]
        write_dataframe(df, str(pix / (fov + ".feather")))
    cell = pd.DataFrame({"fov": g["cell_fov"], "label": g["cell_label"], "cell_size": g["cell_size"]})
    cell_path = str(tmp_path / "cell_table.csv")
    cell.to_csv(cell_path, index=False)
    with warnings.catch_warnings(record=True) as wl:
        warnings.simplefilter("always")
        counts, normed = cell_cluster_utils.create_c2pc_data(fovs, str(pix), cell_path)
    assert [str(w.message) for w in wl if "Pixel clusters" in str(w.message)] == list(g["warnings"])
    for tag, frame in (("counts", counts), ("normed", normed)):
        assert list(frame.columns) == list(g[f"{tag}_columns"])
        assert list(frame["fov"]) == list(g[f"{tag}_fov"])
        np.testing.assert_array_equal(frame.drop(columns="fov").values.astype(np.float64), g[f"{tag}_values"])


def test_tiff_side_percentiles(som_backend, tmp_path):
    """calculate_channel_percentiles / calculate_pixel_intensity_percentile / check_for_modified_channels on
    a small TIFF cohort, against the reference's arithmetic written out with numpy
    (pixel_cluster_utils.py:16-106, :145-181)."""
    from ark_analysis_amd import image_io
    rs = np.random.RandomState(5)
    fovs, chans = ["fov0", "fov1", "fov10"], ["chan10", "chan2", "chan1"]
    images = {}
    for fov in fovs:
        os.makedirs(tmp_path / fov / "TIFs")
        for ch in chans + ["chan2_smoothed"]:
            img = rs.gamma(0.5, 2.0, size=(40, 30)).astype(np.float32)
            img[rs.uniform(size=img.shape) < 0.35] = 0
            if fov == "fov1" and ch == "chan1":
                img[:] = 0                                      # no positive pixel: skipped in the mean
            images[fov, ch] = img
            image_io.write_channel(str(tmp_path / fov / "TIFs" / (ch + ".tiff")), img)
    assert image_io.channel_names(str(tmp_path), "fov0", "TIFs") == ["chan1", "chan2", "chan2_smoothed", "chan10"]
    np.testing.assert_array_equal(image_io.read_channel(str(tmp_path), "fov10", "chan2", "TIFs"), images["fov10", "chan2"])

    got = pixel_cluster_utils.calculate_channel_percentiles(str(tmp_path), fovs, chans, "TIFs", 0.99)
    want = {}
    for ch in chans:
        vals = [np.quantile(images[f, ch][images[f, ch] > 0], 0.99) for f in fovs if (images[f, ch] > 0).any()]
        want[ch] = np.mean(vals)
    assert list(got.columns) == ["chan1", "chan2", "chan10"] and got.shape == (1, 3)
    for ch in chans:
        assert got[ch].dtype == np.float32 and got[ch].values[0] == want[ch]

    thresh = pixel_cluster_utils.calculate_pixel_intensity_percentile(str(tmp_path), fovs, list(got.columns), "TIFs", got)
    per_fov = []
    for f in fovs:
        stack = np.stack([images[f, ch] for ch in got.columns], axis=-1)
        per_fov.append(np.quantile(np.sum(stack / got.iloc[0].values.reshape([1, 1, 3]), axis=-1), 0.05))
    assert thresh == np.mean(per_fov)

    with pytest.warns(UserWarning, match="chan2_smoothed"):
        pixel_cluster_utils.check_for_modified_channels(str(tmp_path), "fov0", "TIFs", ["chan2", "chan1"])


def _write_g9_cohort(g, td):
    from ark_analysis_amd import image_io
    fovs, chans = ["fov0", "fov1", "fov2"], ["chan0", "chan1", "chan2", "chan10"]
    tiff_dir, seg_dir = os.path.join(td, "tiffs"), os.path.join(td, "seg")
    os.makedirs(os.path.join(td, "pixel_output_dir"))
    os.mkdir(seg_dir)
    for fov in fovs:
        os.makedirs(os.path.join(tiff_dir, fov, "TIFs"))
        for ch in chans:
            image_io.write_channel(os.path.join(tiff_dir, fov, "TIFs", ch + ".tiff"), g[f"img_{fov}_{ch}"])
        image_io.write_channel(os.path.join(seg_dir, fov + "_whole_cell.tiff"), g["seg_" + fov])
    return fovs, chans, tiff_dir, seg_dir


def test_create_pixel_matrix_matches_reference_run(som_backend, tmp_path, capsys):
    """TIFF cohort -> pixel tables + the three normalisation files, against the reference's own
    create_pixel_matrix on the same TIFFs (tests/golden/g9_create_pixel_matrix.npz): every value, dtype,
    column order, the seeded sub-sample and the printed progress."""
    from ark_analysis_amd.phenotyping import pixie_preprocessing
    g = np.load(os.path.join(GOLD, "g9_create_pixel_matrix.npz"))
    td = str(tmp_path)
    fovs, chans, tiff_dir, seg_dir = _write_g9_cohort(g, td)
    pixie_preprocessing.create_pixel_matrix(list(fovs), list(chans), td, tiff_dir, seg_dir,
                                            subset_proportion=0.25, seed=42)
    assert capsys.readouterr().out == str(g["stdout"])
    pre = read_dataframe(os.path.join(td, "pixel_output_dir", "channel_norm_pre_rownorm.feather"))
    assert list(pre.columns) == list(g["pre_columns"]) and pre.values.dtype == g["pre_values"].dtype
    np.testing.assert_array_equal(pre.values[0], g["pre_values"])
    th = read_dataframe(os.path.join(td, "pixel_output_dir", "pixel_thresh.feather"))["pixel_thresh_val"].values
    assert th.dtype == g["thresh"].dtype
    np.testing.assert_array_equal(th, g["thresh"])
    post = read_dataframe(os.path.join(td, "channel_norm_post_rownorm.feather"))
    assert list(post.columns) == list(g["post_columns"]) and post.values.dtype == g["post_values"].dtype
    np.testing.assert_array_equal(post.values[0], g["post_values"])
    assert sorted(os.listdir(os.path.join(td, "pixel_mat_data"))) == list(g["data_dir_listing"])
    for fov in fovs:
        for kind in ("pixel_mat_data", "pixel_mat_subsetted"):
            t = read_dataframe(os.path.join(td, kind, fov + ".feather"))
            tag = f"{kind}_{fov}"
            assert list(t.columns) == list(g[tag + "_columns"])
            assert [str(d) for d in t.dtypes] == list(g[tag + "_dtypes"])
            np.testing.assert_array_equal(t[["chan0", "chan1", "chan2", "chan10"]].values, g[tag + "_channels"])
            np.testing.assert_array_equal(t[["row_index", "column_index", "label"]].values.astype(np.int64),
                                          g[tag + "_meta"])
    # nothing left to do on a second call; a changed channel list resets the cohort
    pixie_preprocessing.create_pixel_matrix(list(fovs), list(chans), td, tiff_dir, seg_dir, subset_proportion=0.25)
    assert capsys.readouterr().out == "There are no more FOVs to preprocess, skipping\n"
    pixie_preprocessing.create_pixel_matrix(list(fovs), chans[:3], td, tiff_dir, seg_dir, subset_proportion=0.25)
    out = capsys.readouterr().out
    assert out.startswith("New channels provided: overwriting whole cohort\n") and out.endswith("Processed 3 fovs\n")
    t = read_dataframe(os.path.join(td, "pixel_mat_data", "fov1.feather"))
    assert list(t.columns)[:3] == chans[:3] and "chan10" not in t.columns
    with pytest.raises(ValueError, match="Invalid subset percentage"):
        pixie_preprocessing.create_pixel_matrix(list(fovs), list(chans), td, tiff_dir, seg_dir, subset_proportion=1.5)


def test_channel_stacks_planar_view_prefetch_and_cache(tmp_path):
    """read_channels: same values / dtype / shape as stacking the planes, stored channel-planar; iter_stacks:
    every FOV once and in order, second pass served from the cache while it is under budget."""
    from ark_analysis_amd import image_io
    rs = np.random.RandomState(5)
    fovs, chans = ["fov%d" % i for i in range(4)], ["chan%d" % i for i in range(5)]
    planes = {}
    for fov in fovs:
        os.makedirs(os.path.join(str(tmp_path), fov, "TIFs"))
        for ch in chans:
            planes[fov, ch] = rs.gamma(0.5, 2.0, size=(17, 23)).astype(np.float32)
            image_io.write_channel(os.path.join(str(tmp_path), fov, "TIFs", ch + ".tiff"), planes[fov, ch])
    stack = image_io.read_channels(str(tmp_path), "fov2", chans, "TIFs")
    want = np.stack([planes["fov2", ch] for ch in chans], axis=-1)
    assert stack.shape == (17, 23, 5) and stack.dtype == np.float32
    np.testing.assert_array_equal(stack, want)
    assert stack.transpose(2, 0, 1).flags.c_contiguous           # what flowsom uploads as it lies
    np.testing.assert_array_equal(stack / np.float32(3), want / np.float32(3))
    cache = image_io.stack_cache(max_bytes=2 * stack.nbytes)      # room for two FOVs
    first = [(fov, s.copy()) for fov, s in image_io.iter_stacks(str(tmp_path), fovs, chans, "TIFs", cache=cache)]
    assert [fov for fov, _ in first] == fovs and sorted(k for k in cache if not k.startswith("__")) == fovs[:2]
    os.remove(os.path.join(str(tmp_path), "fov0", "TIFs", "chan0.tiff"))     # cached: not read again
    second = list(image_io.iter_stacks(str(tmp_path), fovs[:2] + fovs[3:], chans, "TIFs", cache=cache, fill=False))
    assert [fov for fov, _ in second] == ["fov0", "fov1", "fov3"]
    for (_, a), (_, b) in zip(second, [first[0], first[1], first[3]]):
        np.testing.assert_array_equal(a, b)
    with pytest.raises(FileNotFoundError):
        list(image_io.iter_stacks(str(tmp_path), ["fov0"], chans, "TIFs"))


@pytest.mark.parametrize("record_name", ["channel_norm_post_rownorm_perfov.csv",
                                         "channel_norm_post_rownorm_perfov.csv.rank1"])
def test_create_pixel_matrix_resumes_after_interruption(som_backend, tmp_path, capsys, record_name):
    """Tables of one FOV missing although its 99.9 % values are on record (a run killed between the two):
    only that FOV is redone and the cohort file comes out as in an uninterrupted run."""
    from ark_analysis_amd.phenotyping import pixie_preprocessing
    g = np.load(os.path.join(GOLD, "g9_create_pixel_matrix.npz"))
    td = str(tmp_path)
    fovs, chans, tiff_dir, seg_dir = _write_g9_cohort(g, td)
    kept = {}
    real_remove = os.remove

    def keep_record(path):            # hold on to the per-FOV record the function deletes at the end
        if path.endswith("channel_norm_post_rownorm_perfov.csv"):
            kept["csv"] = open(path).read()
        real_remove(path)
    os.remove = keep_record
    try:
        pixie_preprocessing.create_pixel_matrix(list(fovs), list(chans), td, tiff_dir, seg_dir,
                                                subset_proportion=0.25, seed=42)
    finally:
        os.remove = real_remove
    capsys.readouterr()
    real_remove(os.path.join(td, "pixel_mat_data", "fov1.feather"))
    real_remove(os.path.join(td, "channel_norm_post_rownorm.feather"))
    # (the record of a single-process run, or what one rank of an interrupted sharded run left behind)
    with open(os.path.join(td, "pixel_mat_data", record_name), "w") as f:
        f.write(kept["csv"])
    pixie_preprocessing.create_pixel_matrix(list(fovs), list(chans), td, tiff_dir, seg_dir,
                                            subset_proportion=0.25, seed=42)
    assert "Restarting preprocessing from FOV fov1, 1 fovs left to process" in capsys.readouterr().out
    post = read_dataframe(os.path.join(td, "channel_norm_post_rownorm.feather"))
    assert list(post.columns) == list(g["post_columns"])
    # the cohort value is a mean over the per-FOV columns in the order they were recorded (set order in the
    # reference, pixie_preprocessing.py:315-327 and :442): a resumed run may differ in the last place
    np.testing.assert_allclose(post.values[0], g["post_values"], rtol=1e-15, atol=0)
    t = read_dataframe(os.path.join(td, "pixel_mat_data", "fov1.feather"))
    np.testing.assert_array_equal(t[["chan0", "chan1", "chan2", "chan10"]].values, g["pixel_mat_data_fov1_channels"])


def test_create_pixel_matrix_record_never_vouches_for_a_failed_table(som_backend, tmp_path, monkeypatch):
    """The tables go out on several writer threads; the per-FOV record (what a restart trusts) lists a FOV only
    once both of its tables -- and those of every FOV before it -- are on disk.  A write that fails stops the
    run, and neither that FOV nor a later one is on record."""
    from ark_analysis_amd import fov_tables
    from ark_analysis_amd.phenotyping import pixie_preprocessing
    g = np.load(os.path.join(GOLD, "g9_create_pixel_matrix.npz"))
    td = str(tmp_path)
    fovs, chans, tiff_dir, seg_dir = _write_g9_cohort(g, td)
    real_write = fov_tables.write_dataframe

    def failing_write(table, path, **kw):
        if path.endswith(os.path.join("pixel_mat_data", "fov1.feather")):
            raise OSError("disk full")
        return real_write(table, path, **kw)
    monkeypatch.setattr(fov_tables, "write_dataframe", failing_write)
    processed = []                 # create_pixel_matrix walks a set: note the order it took
    real_half = pixie_preprocessing._fov_device_half

    def noting_half(*args, **kw):
        processed.append(args[9])
        return real_half(*args, **kw)
    monkeypatch.setattr(pixie_preprocessing, "_fov_device_half", noting_half)
    with pytest.raises(OSError, match="disk full"):
        pixie_preprocessing.create_pixel_matrix(list(fovs), list(chans), td, tiff_dir, seg_dir,
                                                subset_proportion=0.25, seed=42)
    record = os.path.join(td, "pixel_mat_data", "channel_norm_post_rownorm_perfov.csv")
    listed = list(pd.read_csv(record, index_col="channel").columns) if os.path.exists(record) else []
    assert "fov1" not in listed
    for fov in listed:             # whatever is on record has both tables
        assert os.path.exists(os.path.join(td, "pixel_mat_data", fov + ".feather"))
        assert os.path.exists(os.path.join(td, "pixel_mat_subsetted", fov + ".feather"))
    assert all(processed.index(fov) < processed.index("fov1") for fov in listed)
    assert not os.path.exists(os.path.join(td, "channel_norm_post_rownorm.feather"))


def test_generate_pixel_cluster_mask_matches_reference_run(som_backend, tmp_path):
    """Pixel table + mapping -> int16 cluster-id image, against the reference's own function on the same
    inputs (tests/golden/g10_pixel_cluster_mask.npz): SOM and meta columns, float-typed labels, repeated
    mapping rows, ids beyond int8, unlisted pixels left 0; then the error cases."""
    from ark_analysis_amd import image_io
    from ark_analysis_amd.utils import data_utils
    g = np.load(os.path.join(GOLD, "g10_pixel_cluster_mask.npz"))
    td = str(tmp_path)
    h, w = (int(v) for v in g["shape"])
    os.makedirs(os.path.join(td, "tiffs", "fov0"))
    os.makedirs(os.path.join(td, "pixel_mat_data"))
    image_io.write_channel(os.path.join(td, "tiffs", "fov0", "chan0.tiff"), np.zeros((h, w), dtype=np.float32))
    table = pd.DataFrame({"chan0": 0.5, "fov": "fov0", "row_index": g["row_index"], "column_index": g["column_index"],
                          "pixel_som_cluster": g["pixel_som_cluster"], "pixel_meta_cluster": g["pixel_meta_cluster"]})
    write_dataframe(table, os.path.join(td, "pixel_mat_data", "fov0.feather"))
    args = ("fov0", td, os.path.join(td, "tiffs"), os.path.join("fov0", "chan0.tiff"), "pixel_mat_data")
    for col in ("pixel_meta_cluster", "pixel_som_cluster"):
        pairs = g["mapping_" + col]
        mapping = pd.DataFrame({col: pairs[:, 0], "cluster_id": pairs[:, 1], "other": 1.5})
        mask = data_utils.generate_pixel_cluster_mask(*args, mapping, pixel_cluster_col=col)
        assert mask.dtype == np.int16 and mask.shape == (h, w)
        np.testing.assert_array_equal(mask, g["mask_" + col])
    # a pixel listed twice keeps its last row; ids wrap into int16 like numpy's assignment
    twice = pd.concat([table, table.iloc[:3].assign(pixel_som_cluster=2)], ignore_index=True)
    write_dataframe(twice, os.path.join(td, "pixel_mat_data", "fov0.feather"))
    mapping = pd.DataFrame({"pixel_som_cluster": np.arange(1, 13), "cluster_id": np.arange(1, 13) * 7000})
    mask = data_utils.generate_pixel_cluster_mask(*args, mapping, pixel_cluster_col="pixel_som_cluster")
    want = np.zeros(h * w, dtype=np.int16)
    want[twice["row_index"].values * w + twice["column_index"].values] = \
        (twice["pixel_som_cluster"].values * 7000).astype(np.int16)
    np.testing.assert_array_equal(mask, want.reshape(h, w))
    with pytest.raises(KeyError):            # a label the mapping does not know
        data_utils.generate_pixel_cluster_mask(*args, mapping.iloc[:5], pixel_cluster_col="pixel_som_cluster")
    with pytest.raises(ValueError):          # not a cluster column
        data_utils.generate_pixel_cluster_mask(*args, mapping, pixel_cluster_col="fov")
    with pytest.raises(ValueError):          # unknown FOV
        data_utils.generate_pixel_cluster_mask("fov9", *args[1:], mapping, pixel_cluster_col="pixel_som_cluster")
    with pytest.raises(FileNotFoundError):
        data_utils.generate_pixel_cluster_mask("fov0", td, os.path.join(td, "tiffs"), "fov0/none.tiff",
                                               "pixel_mat_data", mapping, pixel_cluster_col="pixel_som_cluster")
    outside = table.assign(row_index=table["row_index"] + h)
    write_dataframe(outside, os.path.join(td, "pixel_mat_data", "fov0.feather"))
    with pytest.raises(IndexError):
        data_utils.generate_pixel_cluster_mask(*args, mapping, pixel_cluster_col="pixel_som_cluster")


def test_saved_pixel_cluster_masks_match_reference_run(som_backend, tmp_path):
    """The cohort loop around the mask (reference utils/data_utils.py:558-635) against a run of the reference on
    the same inputs (tests/golden/g14_saved_pixel_masks.npz): the cluster-name table gets fresh ``cluster_id``s
    (a stale column replaced) and is rewritten in place, one int16 TIFF per FOV lands under the sub-folder."""
    import io
    from ark_analysis_amd import image_io
    from ark_analysis_amd.utils import data_utils
    g = np.load(os.path.join(GOLD, "g14_saved_pixel_masks.npz"))
    td = str(tmp_path)
    h, w = (int(v) for v in g["shape"])
    os.makedirs(os.path.join(td, "pixel_mat_data"))
    os.makedirs(os.path.join(td, "masks"))
    with open(os.path.join(td, "names.csv"), "w") as f:
        f.write(str(g["names_text"][()]))
    for fov in ("fov0", "fov1"):
        os.makedirs(os.path.join(td, "tiffs", fov))
        image_io.write_channel(os.path.join(td, "tiffs", fov, "chan0.tiff"), np.zeros((h, w), dtype=np.float32))
        table = pd.DataFrame({"chan0": 0.5, "fov": fov, "row_index": g["row_index_" + fov],
                              "column_index": g["column_index_" + fov], "pixel_som_cluster": g["som_" + fov],
                              "pixel_meta_cluster": g["meta_" + fov]})
        write_dataframe(table, os.path.join(td, "pixel_mat_data", fov + ".feather"))
    data_utils.generate_and_save_pixel_cluster_masks(
        ["fov0", "fov1"], td, os.path.join(td, "masks"), os.path.join(td, "tiffs"), "chan0.tiff", "pixel_mat_data",
        os.path.join(td, "names.csv"), pixel_cluster_col="pixel_meta_cluster", sub_dir="pixel_masks",
        name_suffix="_pixel_mask")
    got = pd.read_csv(os.path.join(td, "names.csv"))
    want = pd.read_csv(io.StringIO(str(g["names_after_text"][()])))
    pd.testing.assert_frame_equal(got, want)
    for fov in ("fov0", "fov1"):
        mask = image_io.read_image(os.path.join(td, "masks", "pixel_masks", fov + "_pixel_mask.tiff"))
        assert mask.dtype == np.int16 and mask.shape == (h, w)
        np.testing.assert_array_equal(mask, g["mask_" + fov])
    # without a sub-folder the files go straight into save_dir; a missing save_dir is an error
    data_utils.save_fov_mask("fovA", os.path.join(td, "masks"), np.arange(6, dtype=np.int16).reshape(2, 3))
    np.testing.assert_array_equal(image_io.read_image(os.path.join(td, "masks", "fovA.tiff")),
                                  np.arange(6).reshape(2, 3))
    with pytest.raises(FileNotFoundError):
        data_utils.save_fov_mask("fovA", os.path.join(td, "nowhere"), np.zeros((2, 2), dtype=np.int16))
    # mask dtypes the reference's tifffile accepts and the baseline writer has no sample format for: same pixel values
    for mask in (np.array([[0, 70000], [3, 1]], dtype=np.uint32), np.arange(6, dtype=np.int64).reshape(2, 3) - 2,
                 np.array([[True, False]]), np.array([[0.5, 2.0]], dtype=np.float64)):
        data_utils.save_fov_mask("fovB", os.path.join(td, "masks"), mask)
        np.testing.assert_array_equal(image_io.read_image(os.path.join(td, "masks", "fovB.tiff")), mask)
    with pytest.raises(ValueError):
        data_utils.save_fov_mask("fovB", os.path.join(td, "masks"), np.array([[2 ** 40]], dtype=np.int64))
    # the TIFF writer keeps every dtype it accepts; Pillow agrees on the values
    from PIL import Image
    for dtype in (np.uint8, np.uint16, np.int16, np.int32, np.float32):
        image = (np.random.RandomState(3).randn(9, 4) * 90).astype(dtype)
        image_io.write_image(os.path.join(td, "rt.tiff"), image)
        back = image_io.read_image(os.path.join(td, "rt.tiff"))
        assert back.dtype == image.dtype
        np.testing.assert_array_equal(back, image)
        with Image.open(os.path.join(td, "rt.tiff")) as im:
            np.testing.assert_array_equal(np.array(im), image)
    for bad in (np.zeros((2, 2, 2), dtype=np.int16), np.zeros((2, 2), dtype=np.float64), np.zeros((2, 2), dtype=bool)):
        with pytest.raises(ValueError):
            image_io.write_image(os.path.join(td, "rt.tiff"), bad)


def test_fov_table_helpers(tmp_path):
    """Footer-only column listing, natural ordering, prefetcher / writer round trip, damaged files."""
    from ark_analysis_amd.fov_tables import FovTableDir, TablePrefetcher, TableWriter
    root = tmp_path / "tabs"
    root.mkdir()
    names = ["fov10", "fov2", "fov1"]
    frames = {n: pd.DataFrame({"a": np.arange(5.0) + i, "segmentation_label": np.arange(5)})
              for i, n in enumerate(names)}
    w = TableWriter()
    for n, df in frames.items():
        w.submit(df, str(root / (n + ".feather")))
    w.close()
    (root / "fov3.feather").write_text("not an arrow file")
    (root / ".hidden.feather").write_text("x")
    tabs = FovTableDir(str(root))
    assert tabs.fovs() == ["fov1", "fov2", "fov3", "fov10"]
    assert tabs.column_names("fov2") == ["a", "segmentation_label"]
    got = list(TablePrefetcher(tabs, tabs.fovs(), depth=1))
    assert [f for f, _ in got] == tabs.fovs()
    assert got[2][1] is None                                  # damaged table
    for fov, table in got:
        if table is not None:
            pd.testing.assert_frame_equal(table, frames[fov])
    with pytest.raises(OSError):
        bad = TableWriter()
        bad.submit(frames["fov1"], str(tmp_path / "no_such_dir" / "x.feather"))
        bad.close()
    # several readers still deliver in FOV order; several writers report every table as finished (written or not)
    got = list(TablePrefetcher(tabs, tabs.fovs(), depth=3, workers=3))
    assert [f for f, _ in got] == tabs.fovs() and got[2][1] is None
    finished = []
    w = TableWriter(depth=4, workers=3)
    for n, df in frames.items():
        w.submit(df, str(root / (n + "_copy.feather")), done=lambda n=n: finished.append(n))
    w.close()
    assert sorted(finished) == sorted(frames)
    # commit: the staging twin replaces the directory at once, the old tables disappear in the background
    from ark_analysis_amd import fov_tables
    tabs.open_staging()
    TableWriter().close()
    fov_tables.write_dataframe(frames["fov1"], tabs.path("only", staged=True))
    tabs.commit()
    assert tabs.fovs() == ["only"] and not os.path.exists(tabs.staging)
    fov_tables.wait_for_cleanup()
    assert sorted(os.listdir(tmp_path)) == ["tabs"]


def test_raw_tiff_reader_equals_pillow(tmp_path):
    """image_io.read_image: uncompressed strips read straight from the file == np.array(PIL.Image.open(path)) for
    every sample format the pixel path meets; compressed files and short files take Pillow's route."""
    from PIL import Image
    from ark_analysis_amd import image_io
    rs = np.random.RandomState(3)
    for dt in (np.float32, np.int32, np.uint16, np.uint8, np.int16, np.float64):
        img = (rs.rand(37, 53) * 1000).astype(dt)
        path = str(tmp_path / "img.tiff")
        Image.fromarray(img).save(path, format="TIFF")
        with Image.open(path) as im:
            assert image_io._raw_layout(im) is not None       # (before the pixels are loaded: Pillow drops the tiles)
            want = np.array(im)
        got = image_io.read_image(path)
        assert got.dtype == want.dtype
        np.testing.assert_array_equal(got, want)
        out = np.empty(want.shape, dtype=want.dtype)
        assert image_io.read_image(path, out=out) is out
        np.testing.assert_array_equal(out, want)
        wrong = np.empty(want.shape, dtype=np.complex64)          # unusable destination: a new array comes back
        np.testing.assert_array_equal(image_io.read_image(path, out=wrong), want)
    img = rs.rand(64, 64).astype(np.float32)
    packed = str(tmp_path / "deflate.tiff")
    Image.fromarray(img).save(packed, format="TIFF", compression="tiff_deflate")
    with Image.open(packed) as im:
        assert image_io._raw_layout(im) is None
    np.testing.assert_array_equal(image_io.read_image(packed), img)
    whole = str(tmp_path / "whole.tiff")
    Image.fromarray(img).save(whole, format="TIFF")
    data = open(whole, "rb").read()
    with open(whole, "wb") as f:
        f.write(data[:len(data) // 2])
    with pytest.raises(OSError):
        image_io.read_image(whole)


def test_host_blocks_are_recycled():
    from ark_analysis_amd.arrow_assign import HostBlocks
    pool = HostBlocks()
    a = pool.take(1000)
    b = pool.take(500)
    assert a.numel() >= 1000 and b.numel() >= 500 and a.data_ptr() != b.data_ptr()
    pool.give(a)
    pool.give(b)
    assert pool.take(400).data_ptr() == b.data_ptr()      # the smallest block that fits
    assert pool.take(400).data_ptr() == a.data_ptr()
    assert pool.take(2000).numel() >= 2000                # nothing fits: a new block
    pool.close()


def test_reference_generate_som_avg_files(som_backend, tmp_path, capsys):
    g = np.load(os.path.join(GOLD, "g7_pixel_pipeline.npz"))
    td = str(tmp_path)
    _build_pixel_dirs(td, g)
    obj = pixel_som_clustering.train_pixel_som(FOVS, CHANS, td)
    pixel_som_clustering.cluster_pixels(FOVS, td, obj)
    capsys.readouterr()
    pixel_som_clustering.generate_som_avg_files(FOVS, CHANS, td, obj, data_dir="pixel_mat_data")
    assert capsys.readouterr().out == "Computing average channel expression across pixel SOM clusters\n"
    pixel_som_clustering.generate_som_avg_files(FOVS, CHANS, td, obj, data_dir="pixel_mat_data")
    assert capsys.readouterr().out == "Already generated SOM cluster channel average file, skipping\n"
    pixel_som_clustering.generate_som_avg_files(FOVS, CHANS, td, obj, data_dir="pixel_mat_data", overwrite=True)
    assert capsys.readouterr().out.startswith(
        "Overwrite flag set, regenerating SOM cluster channel average file\n")
    # clusters lost by FOV sub-sampling -> ValueError, file not written
    obj.som_clusters_seen = set(range(1, 151))
    os.remove(os.path.join(td, "pixel_channel_avg_som_cluster.csv"))
    with pytest.raises(ValueError, match="Average expression file not written"):
        pixel_som_clustering.generate_som_avg_files(FOVS, CHANS, td, obj, data_dir="pixel_mat_data")
    with warnings.catch_warnings(record=True) as wlist:
        warnings.simplefilter("always")
        pixel_cluster_utils.compute_pixel_cluster_channel_avg(FOVS, CHANS, td, "pixel_som_cluster", None,
                                                              "pixel_mat_data", num_fovs_subset=100)
    assert any("Provided num_fovs_subset" in str(w.message) for w in wlist)


@pytest.mark.gpu
def test_avg_files_use_the_totals_cluster_pixels_left_behind(tmp_path, monkeypatch):
    """generate_som_avg_files right after cluster_pixels reads no table again (the per-cluster totals were taken
    while the rows were in HBM), gives the numbers a fresh pass over the files gives, and falls back to reading as
    soon as a file is no longer the one that was written."""
    from ark_analysis_amd import fov_tables
    g = np.load(os.path.join(GOLD, "g7_pixel_pipeline.npz"))
    td = str(tmp_path)
    _build_pixel_dirs(td, g)
    obj = pixel_som_clustering.train_pixel_som(FOVS, CHANS, td)
    pixel_som_clustering.cluster_pixels(FOVS, td, obj)
    reads = []
    real = fov_tables.FovTableDir.load_arrow
    monkeypatch.setattr(fov_tables.FovTableDir, "load_arrow", lambda self, fov: reads.append(fov) or real(self, fov))
    pixel_som_clustering.generate_som_avg_files(FOVS, CHANS, td, obj, data_dir="pixel_mat_data")
    assert reads == []
    cached = pd.read_csv(os.path.join(td, "pixel_channel_avg_som_cluster.csv"))
    fresh = pixel_cluster_utils.compute_pixel_cluster_channel_avg(FOVS, CHANS, td, "pixel_som_cluster", None,
                                                                  "pixel_mat_data", num_fovs_subset=len(FOVS), keep_count=True)
    assert sorted(reads) == sorted(FOVS)
    np.testing.assert_array_equal(cached["count"].values, fresh["count"].values)
    np.testing.assert_allclose(cached[CHANS].values, fresh[CHANS].values, rtol=1e-13, atol=0)
    # a rewritten table (same content, new modification time) is read again
    del reads[:]
    path = os.path.join(td, "pixel_mat_data", FOVS[0] + ".feather")
    os.utime(path, ns=(1, 1))
    pixel_som_clustering.generate_som_avg_files(FOVS, CHANS, td, obj, data_dir="pixel_mat_data", overwrite=True)
    assert reads == [FOVS[0]]
