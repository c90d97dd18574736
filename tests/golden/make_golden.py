#!/usr/bin/env python
"""Generates the golden fixtures in tests/golden/*.npz by importing the REFERENCE
(/root/reference/src, Python) in the build container, with tests/golden/_shims standing in for
the third-party packages this image lacks (feather, natsort, alpineer, skimage, pyFlowSOM).

Only small input/output arrays are written -- no reference source travels.  The pyFlowSOM shim is
backed by the build's CPU oracle, so fixtures that cross a SOM call (g6_*, g7_*) carry
"oracle-of-record = build restatement"; everything else is pure reference numerics
(numpy / scipy / pandas as the reference calls them).

    python tests/golden/make_golden.py        (needs /root/reference; never runs on the GPU box)
"""
import io
import os
import sys
import tempfile
import contextlib

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(HERE, "_shims"))
sys.path.insert(0, "/root/reference/src")

import feather  # noqa: E402  (shim)
from ark.phenotyping import (cluster_helpers, pixel_cluster_utils, pixel_som_clustering,  # noqa: E402
                             pixie_preprocessing, cell_som_clustering)
import scipy.ndimage as ndimage  # noqa: E402

from tests import oracle_binding as ob  # noqa: E402


OUT_DIR = os.environ.get("PXSOM_GOLDEN_OUT", HERE)     # tests/test_golden_regenerates.py writes to a scratch directory


def save(name, **arrays):
    path = os.path.join(OUT_DIR, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote", os.path.relpath(path, ROOT), {k: getattr(v, "shape", None) for k, v in arrays.items()})


def g1_normalize():
    rs = np.random.RandomState(1)
    chans = ["chan%d" % i for i in range(6)]
    with tempfile.TemporaryDirectory() as td:
        sub = os.path.join(td, "sub")
        os.mkdir(sub)
        norm = pd.DataFrame(rs.uniform(0.2, 1.5, size=(1, 6)), columns=chans)
        feather.write_dataframe(norm, os.path.join(td, "norm.feather"))
        df = pd.DataFrame(rs.rand(50, 6), columns=chans)
        df["fov"] = "fov0"
        feather.write_dataframe(df, os.path.join(sub, "fov0.feather"))
        obj = cluster_helpers.PixelSOMCluster(sub, os.path.join(td, "norm.feather"),
                                              os.path.join(td, "w.feather"), ["fov0"], chans)
        ext = pd.DataFrame(rs.rand(200, 6) * 3, columns=chans)
        ext.iloc[::7, 2] = 0.0
        out = obj.normalize_data(ext)
    save("g1_normalize", x=ext.values, norm=norm.values[0], out=out[chans].values)


def g2_g5_preprocess():
    for tag, (h, w, c, seed, thresh) in {"a": (32, 32, 3, 2, 0.0), "b": (24, 40, 8, 3, 0.5),
                                         "c": (32, 32, 22, 4, 3.0)}.items():
        rs = np.random.RandomState(seed)
        img = rs.gamma(0.6, 1.0, size=(h, w, c))
        img[rs.rand(h, w, c) < 0.35] = 0.0        # MIBI-like sparsity
        img[:3, :3, :] = 0.0                      # a few all-zero pixels
        chans = ["chan%d" % i for i in range(c)]
        # G5: the blur exactly as pixie_preprocessing.py:47-49 calls it
        blurred = np.stack([ndimage.gaussian_filter(img[:, :, i], sigma=2) for i in range(c)], axis=-1)
        np.random.seed(7)
        full, sub = pixie_preprocessing.create_fov_pixel_data(
            "fov0", list(chans), img.copy(), None, pixel_thresh_val=thresh)
        kept = (full["row_index"].values * w + full["column_index"].values).astype(np.int64)
        save("g2_fovpixel_" + tag, img=img, blurred=blurred, thresh=np.float64(thresh),
             kept_index=kept, rows=full[chans].values, subset_len=np.int64(len(sub)))
        # the dtype the pipeline really feeds (preprocess_fov: float32 TIFF / float32 norm values): scipy and
        # pandas then stay in float32
        img32 = img.astype(np.float32)
        blurred32 = np.stack([ndimage.gaussian_filter(img32[:, :, i], sigma=2) for i in range(c)], axis=-1)
        np.random.seed(7)
        full32, sub32 = pixie_preprocessing.create_fov_pixel_data(
            "fov0", list(chans), img32.copy(), None, pixel_thresh_val=np.float32(thresh))
        assert blurred32.dtype == np.float32 and full32[chans].values.dtype == np.float32
        kept32 = (full32["row_index"].values * w + full32["column_index"].values).astype(np.int64)
        save("g2_fovpixel_" + tag + "_f32", img=img32, blurred=blurred32, thresh=np.float32(thresh),
             kept_index=kept32, rows=full32[chans].values, subset_index=sub32.index.values.astype(np.int64))


def g3_quantiles():
    rs = np.random.RandomState(5)
    cases = {}
    for i, n in enumerate([1, 2, 3, 10, 999, 1000, 1001, 5003]):
        col = rs.gamma(0.5, 1.0, size=n)
        col[rs.rand(n) < 0.3] = 0.0
        if i == 1:
            col[:] = 0.0        # all-zero column -> NaN
        df = pd.DataFrame({"c": col})
        cases["x%d" % i] = col
        cases["q999_%d" % i] = np.float64(df.replace(0, np.nan).quantile(q=0.999, axis=0)["c"])
        pos = col[col > 0]
        cases["q99pos_%d" % i] = np.float64(np.quantile(pos, 0.99)) if len(pos) else np.float64(np.nan)
        cases["q05_%d" % i] = np.float64(np.quantile(col, 0.05))
    save("g3_quantiles", **cases)


def g4_cluster_avg():
    rs = np.random.RandomState(6)
    chans = ["chan%d" % i for i in range(5)]
    with tempfile.TemporaryDirectory() as td:
        os.mkdir(os.path.join(td, "pixel_mat_data"))
        arrays = {}
        fovs = ["fov%d" % i for i in range(4)]
        for i, fov in enumerate(fovs):
            n = 300 + 17 * i
            df = pd.DataFrame(rs.rand(n, 5), columns=chans)
            df["fov"] = fov
            df["pixel_som_cluster"] = rs.randint(1, 12, size=n)
            feather.write_dataframe(df, os.path.join(td, "pixel_mat_data", fov + ".feather"))
            arrays["x_" + fov] = df[chans].values
            arrays["lab_" + fov] = df["pixel_som_cluster"].values.astype(np.int64)
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out = pixel_cluster_utils.compute_pixel_cluster_channel_avg(
                fovs, chans, td, "pixel_som_cluster", None, "pixel_mat_data", num_fovs_subset=100,
                seed=42, keep_count=True)
            out3 = pixel_cluster_utils.compute_pixel_cluster_channel_avg(
                fovs, chans, td, "pixel_som_cluster", None, "pixel_mat_data", num_fovs_subset=3,
                seed=42, keep_count=True)
    save("g4_cluster_avg", clusters=out["pixel_som_cluster"].values.astype(np.int64),
         means=out[chans].values, count=out["count"].values.astype(np.int64),
         means_sub3=out3[chans].values, count_sub3=out3["count"].values.astype(np.int64), **arrays)


def g5_meta_clustering():
    """The reference's own meta-clustering step: PixieConsensusCluster (z-score, cap, Ward cut) on a seeded 100 x 6
    table, then pixel_consensus_cluster -> generate_meta_avg_files -> apply_pixel_meta_cluster_remapping ->
    generate_remap_avg_files on three small FOV tables (SURVEY.md section 8(c) "G5")."""
    from ark.phenotyping import pixel_meta_clustering
    rs = np.random.RandomState(5)
    arrays = {}
    cols = ["m%d" % i for i in range(6)]
    with tempfile.TemporaryDirectory() as td:
        table = pd.DataFrame(rs.gamma(1.2, 0.7, size=(100, 6)), columns=cols)
        table.iloc[:, 3] *= 40.0                     # a column whose z-scores reach the cap
        table.insert(0, "pixel_som_cluster", np.arange(1, 101))
        table["count"] = rs.randint(10, 1000, size=100)
        path = os.path.join(td, "avg.csv")
        table.to_csv(path, index=False)
        cc = cluster_helpers.PixieConsensusCluster("pixel", path, cols, max_k=20, cap=3)
        cc.scale_data()
        np.random.seed(42)
        cc.run_consensus_clustering()
        cc.generate_som_to_meta_map()
        arrays.update(ward_table=table[cols].values, ward_scaled=cc.input_data[cols].values,
                      ward_mapping=cc.mapping.values.astype(np.int64), ward_bestk=np.int64(cc.cc.bestK),
                      ward_mk_shape=np.array(cc.cc.Mk.shape, dtype=np.int64))

    chans = ["chan%d" % i for i in range(4)]
    fovs = ["fov0", "fov1", "fov2"]
    with tempfile.TemporaryDirectory() as td:
        os.mkdir(os.path.join(td, "pixel_mat_data"))
        centers = rs.rand(30, 4)
        for fov in fovs:
            n = 1200
            som = rs.randint(1, 31, size=n)
            x = np.abs(centers[som - 1] + 0.05 * rs.standard_normal((n, 4)))
            df = pd.DataFrame(x, columns=chans)
            df["fov"] = fov
            df["row_index"] = np.repeat(np.arange(n // 40), 40)
            df["column_index"] = np.tile(np.arange(40), n // 40)
            df["label"] = rs.randint(0, 20, size=n)
            df["pixel_som_cluster"] = som
            feather.write_dataframe(df, os.path.join(td, "pixel_mat_data", fov + ".feather"))
            arrays["data_" + fov], arrays["som_" + fov] = x, som.astype(np.int64)
            arrays["meta_" + fov] = df[["row_index", "column_index", "label"]].values.astype(np.int64)
        som_avg = pixel_cluster_utils.compute_pixel_cluster_channel_avg(
            fovs, chans, td, "pixel_som_cluster", 30, "pixel_mat_data", keep_count=True)
        som_avg.to_csv(os.path.join(td, "pixel_channel_avg_som_cluster.csv"), index=False)
        arrays["som_avg"] = som_avg.values.astype(np.float64)
        arrays["som_avg_columns"] = np.array(list(som_avg.columns), dtype="U32")
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            pcc = pixel_meta_clustering.pixel_consensus_cluster(fovs, chans, td, max_k=6, cap=3)
            pixel_meta_clustering.generate_meta_avg_files(fovs, chans, td, pcc)
        arrays["stdout_consensus"] = np.array(buf.getvalue())
        arrays["mapping"] = pcc.mapping.values.astype(np.int64)
        for fov in fovs:
            res = feather.read_dataframe(os.path.join(td, "pixel_mat_data", fov + ".feather"))
            arrays["columns_after_consensus"] = np.array(list(res.columns), dtype="U32")
            arrays["metalab_" + fov] = res["pixel_meta_cluster"].values.astype(np.int64)
        meta_avg = pd.read_csv(os.path.join(td, "pixel_channel_avg_meta_cluster.csv"))
        som_avg2 = pd.read_csv(os.path.join(td, "pixel_channel_avg_som_cluster.csv"))
        arrays.update(meta_avg=meta_avg.values.astype(np.float64), meta_avg_columns=np.array(list(meta_avg.columns), dtype="U32"),
                      som_avg_after=som_avg2.values.astype(np.float64),
                      som_avg_after_columns=np.array(list(som_avg2.columns), dtype="U32"))
        # a manual remapping: meta clusters 5 and 6 merged into 5, everything named
        remap = pcc.mapping.copy()
        remap.loc[remap["pixel_meta_cluster"] == 6, "pixel_meta_cluster"] = 5
        names = {1: "tumor", 2: "CD4 T", 3: "stroma_1", 4: "B", 5: "other"}
        remap["pixel_meta_cluster_rename"] = remap["pixel_meta_cluster"].map(names)
        remap.to_csv(os.path.join(td, "remap.csv"), index=False)
        arrays["remap_som"] = remap["pixel_som_cluster"].values.astype(np.int64)
        arrays["remap_meta"] = remap["pixel_meta_cluster"].values.astype(np.int64)
        arrays["remap_name"] = remap["pixel_meta_cluster_rename"].values.astype("U16")
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            pixel_meta_clustering.apply_pixel_meta_cluster_remapping(fovs, chans, td, "pixel_mat_data", "remap.csv")
            pixel_meta_clustering.generate_remap_avg_files(
                fovs, chans, td, "pixel_mat_data", "remap.csv", "pixel_channel_avg_som_cluster.csv",
                "pixel_channel_avg_meta_cluster.csv")
        arrays["stdout_remap"] = np.array(buf.getvalue())
        for fov in fovs:
            res = feather.read_dataframe(os.path.join(td, "pixel_mat_data", fov + ".feather"))
            arrays["columns_after_remap"] = np.array(list(res.columns), dtype="U32")
            arrays["dtypes_after_remap"] = np.array([str(t) for t in res.dtypes], dtype="U16")
            arrays["remapped_" + fov] = res["pixel_meta_cluster"].values.astype(np.int64)
            arrays["renamed_" + fov] = res["pixel_meta_cluster_rename"].values.astype("U16")
        meta_avg3 = pd.read_csv(os.path.join(td, "pixel_channel_avg_meta_cluster.csv"))
        som_avg3 = pd.read_csv(os.path.join(td, "pixel_channel_avg_som_cluster.csv"))
        num = [c for c in meta_avg3.columns if c != "pixel_meta_cluster_rename"]
        arrays.update(meta_avg_remap=meta_avg3[num].values.astype(np.float64),
                      meta_avg_remap_columns=np.array(list(meta_avg3.columns), dtype="U32"),
                      meta_avg_remap_names=meta_avg3["pixel_meta_cluster_rename"].values.astype("U16"),
                      som_avg_remap_columns=np.array(list(som_avg3.columns), dtype="U32"),
                      som_avg_remap_meta=som_avg3["pixel_meta_cluster"].values.astype(np.int64),
                      som_avg_remap_names=som_avg3["pixel_meta_cluster_rename"].values.astype("U16"))
    save("g5_meta_clustering", **arrays)


def g6_som():
    """Oracle-of-record SOM vectors (pyFlowSOM itself is absent: parity unpinned)."""
    from ark_analysis_amd.flowsom import default_radius_range
    out = {}
    for tag, (n, c, xd, yd, rlen, seed) in {"a": (600, 4, 10, 10, 1, 1), "b": (900, 22, 10, 10, 2, 2),
                                            "c": (500, 15, 20, 10, 1, 3), "d": (450, 40, 20, 20, 1, 4)}.items():
        rs = np.random.RandomState(seed)
        x = rs.gamma(0.7, 0.4, size=(n, c))
        x[rs.rand(n, c) < 0.2] = 0
        k = xd * yd
        init = x[rs.choice(n, k, replace=False)].copy()
        order = rs.randint(0, n, size=n * rlen).astype(np.int64)
        rr = default_radius_range(xd, yd)
        w = ob.som_online(x, init, xd, yd, rlen, (0.05, 0.01), rr, order)
        w[k - 1] = w[k // 2]                       # an exact tie pair for the assignment test
        test = np.concatenate([x, w[[0, k // 2, k - 1]], 0.5 * (w[3:4] + w[4:5])])
        labels, dists = ob.map_data_to_nodes(w, test)
        wb = ob.som_batch(x, init, xd, yd, rlen, (0.05, 0.01), rr, 8)
        out.update({f"{tag}_x": x, f"{tag}_init": init, f"{tag}_order": order,
                    f"{tag}_grid": np.array([xd, yd, rlen], dtype=np.int64), f"{tag}_w": w,
                    f"{tag}_test": test, f"{tag}_labels": labels, f"{tag}_dists": dists,
                    f"{tag}_wbatch8": wb})
    save("g6_som_oracle", **out)


def g7b_batch_mode():
    """g7's pixel pipeline once more with the shim's som() running the build's BATCH rule (8 mini-batch steps):
    the fixture of train_pixel_som(..., train_mode="batch", batch_steps=8)."""
    os.environ["PXSOM_SHIM_BATCH_STEPS"] = "8"
    try:
        _g7_pixel("g7b_pixel_pipeline_batch")
    finally:
        del os.environ["PXSOM_SHIM_BATCH_STEPS"]


def g7_end_to_end():
    """The reference's own train_pixel_som -> cluster_pixels -> generate_som_avg_files, with the
    pyFlowSOM shim (oracle) underneath: fixtures for the drop-in pipeline functions."""
    _g7_pixel("g7_pixel_pipeline")
    _g7_cell()


def _g7_pixel(name):
    rs = np.random.RandomState(11)
    chans = ["chan%d" % i for i in range(4)]
    fovs = ["fov0", "fov1", "fov2"]
    arrays = {}
    with tempfile.TemporaryDirectory() as td:
        os.mkdir(os.path.join(td, "pixel_mat_data"))
        os.mkdir(os.path.join(td, "pixel_mat_subsetted"))
        norm = pd.DataFrame(rs.uniform(0.3, 0.9, size=(1, 4)), columns=chans)
        feather.write_dataframe(norm, os.path.join(td, "post_rowsum_chan_norm.feather"))
        arrays["norm"] = norm.values[0]
        for fov in fovs:
            n = 1500
            x = rs.gamma(0.8, 0.3, size=(n, 4))
            x = x / x.sum(axis=1, keepdims=True)
            df = pd.DataFrame(x, columns=chans)
            df["fov"] = fov
            df["row_index"] = np.repeat(np.arange(n // 50), 50)
            df["column_index"] = np.tile(np.arange(50), n // 50)
            df["label"] = rs.randint(0, 30, size=n)
            feather.write_dataframe(df, os.path.join(td, "pixel_mat_data", fov + ".feather"))
            sub = df.iloc[rs.choice(n, 400, replace=False)]
            feather.write_dataframe(sub, os.path.join(td, "pixel_mat_subsetted", fov + ".feather"))
            arrays["data_" + fov] = x
            arrays["meta_" + fov] = df[["row_index", "column_index", "label"]].values.astype(np.int64)
            arrays["sub_" + fov] = sub[chans].values
            arrays["subidx_" + fov] = sub.index.values.astype(np.int64)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            obj = pixel_som_clustering.train_pixel_som(fovs, chans, td, num_passes=1, seed=42)
            pixel_som_clustering.cluster_pixels(fovs, td, obj)
            pixel_som_clustering.generate_som_avg_files(fovs, chans, td, obj, data_dir="pixel_mat_data")
        arrays["stdout"] = np.array(buf.getvalue())
        arrays["weights"] = obj.weights.values
        for fov in fovs:
            res = feather.read_dataframe(os.path.join(td, "pixel_mat_data", fov + ".feather"))
            arrays["labels_" + fov] = res["pixel_som_cluster"].values.astype(np.int64)
            arrays["normed_" + fov] = res[chans].values
        avg = pd.read_csv(os.path.join(td, "pixel_channel_avg_som_cluster.csv"))
        arrays["avg_clusters"] = avg["pixel_som_cluster"].values.astype(np.int64)
        arrays["avg_means"] = avg[chans].values
        arrays["avg_count"] = avg["count"].values.astype(np.int64)
        arrays["clusters_seen"] = np.array(sorted(int(v) for v in obj.som_clusters_seen), dtype=np.int64)
    save(name, **arrays)


def _g7_cell():
    # cell path (reference: cell_som_clustering.py + CellSOMCluster 99.9 % normalisation)
    rs = np.random.RandomState(12)
    cols = ["pixel_meta_cluster_%d" % i for i in range(1, 9)]
    n = 1200
    cell = pd.DataFrame(rs.poisson(3.0, size=(n, 8)) / rs.uniform(50, 500, size=(n, 1)), columns=cols)
    cell["fov"] = rs.choice(["fov0", "fov1"], size=n)
    cell["segmentation_label"] = np.arange(n)
    cell["cell_size"] = rs.randint(50, 500, size=n)
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "cell_table.csv"), "w").write("x\n")
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            cobj = cell_som_clustering.train_cell_som(["fov0", "fov1"], td, os.path.join(td, "cell_table.csv"),
                                                      cols, cell.copy(), seed=42)
            res = cell_som_clustering.cluster_cells(td, cobj, cols)
    save("g7_cell_pipeline", cell=cell[cols].values, fov=cell["fov"].values.astype("U8"),
         cell_size=cell["cell_size"].values.astype(np.int64), weights=cobj.weights.values,
         normed=res[cols].values, labels=res["cell_som_cluster"].values.astype(np.int64),
         stdout=np.array(buf.getvalue()))


def g8_c2pc():
    """The reference's create_c2pc_data (cell x pixel-cluster count matrix + its cell_size-normalised
    twin) on three small FOV tables: clusters that never fall inside a listed cell (dropped with a
    warning), a float-typed cluster column, cells without pixels, background label 0, a FOV of the cell
    table that is not requested."""
    import warnings
    from ark.phenotyping import cell_cluster_utils
    rs = np.random.RandomState(21)
    fovs = ["fov0", "fov1", "fov2"]
    arrays = {}
    with tempfile.TemporaryDirectory() as td:
        pix = os.path.join(td, "pixel_mat_data")
        os.mkdir(pix)
        for i, fov in enumerate(fovs):
            n = 900
            df = pd.DataFrame({"chan0": rs.rand(n)})
            df["fov"] = fov
            df["row_index"] = np.repeat(np.arange(30), 30)
            df["column_index"] = np.tile(np.arange(30), 30)
            lab = rs.randint(0, 26, size=n)
            som = rs.randint(1, 13, size=n)
            meta = rs.randint(1, 7, size=n)
            meta[lab > 0] = np.where(meta[lab > 0] == 6, 5, meta[lab > 0])   # cluster 6 only on background
            som[(som == 12) & (lab != 25)] = 11                              # cluster 12 only in cell 25
            df["segmentation_label" if i == 1 else "label"] = lab
            df["pixel_som_cluster"] = som
            df["pixel_meta_cluster_rename"] = meta.astype(np.float64) if i == 2 else meta
            feather.write_dataframe(df, os.path.join(pix, fov + ".feather"))
            arrays["lab_" + fov], arrays["som_" + fov], arrays["meta_" + fov] = lab, som, meta
        rows = []
        for fov in fovs + ["fov9"]:
            for lab in range(1, 25):             # cell 25 is not in the cell table, cells 1..24 are
                rows.append((fov, lab, int(rs.randint(20, 200)), rs.rand()))
        rows.append(("fov0", 40, 77, 0.5))       # a cell without pixels
        cell = pd.DataFrame(rows, columns=["fov", "label", "cell_size", "extra"])
        # FOVs interleaved, labels ascending within a FOV (as ark's own cell tables are).  NB: the reference
        # pairs count rows (in set-iteration order of the labels) with cell rows (in table order)
        # positionally, cell_cluster_utils.py:152-166 -- for a table not sorted by label it attaches
        # counts to the wrong cells; the fixture stays inside the well-defined case.
        cell = cell.sample(frac=1.0, random_state=3).sort_values("label", kind="stable").reset_index(drop=True)
        cell_path = os.path.join(td, "cell_table.csv")
        cell.to_csv(cell_path, index=False)
        arrays["cell_fov"] = cell["fov"].values.astype("U8")
        arrays["cell_label"] = cell["label"].values.astype(np.int64)
        arrays["cell_size"] = cell["cell_size"].values.astype(np.int64)
        arrays["cell_extra"] = cell["extra"].values
        for col in ("pixel_som_cluster", "pixel_meta_cluster_rename"):
            with warnings.catch_warnings(record=True) as wl:
                warnings.simplefilter("always")
                counts, normed = cell_cluster_utils.create_c2pc_data(fovs, pix, cell_path, col)
            arrays[col + "_warnings"] = np.array([str(w.message) for w in wl
                                                  if not str(w.message).startswith("pyarrow.feather")], dtype="U300")
            for tag, frame in (("counts", counts), ("normed", normed)):
                arrays[f"{col}_{tag}_columns"] = np.array(list(frame.columns), dtype="U64")
                arrays[f"{col}_{tag}_dtypes"] = np.array([str(t) for t in frame.dtypes], dtype="U16")
                arrays[f"{col}_{tag}_fov"] = frame["fov"].values.astype("U8")
                arrays[f"{col}_{tag}_values"] = frame.drop(columns="fov").values.astype(np.float64)
    save("g8_c2pc", **arrays)


def g8s_c2pc_named():
    """create_c2pc_data once more with the default column holding the free-text names the meta-cluster remapping
    step assigns ('CD4_T', 'tumor 2', ...): the reference uses them as they are, in the pivot and in the column
    names (cell_cluster_utils.py:128-146)."""
    import warnings
    from ark.phenotyping import cell_cluster_utils
    rs = np.random.RandomState(22)
    names = np.array(["CD4_T", "B cell", "tumor 2", "10", "2", "stroma", "only_background"])
    fovs = ["fov0", "fov1"]
    arrays = {"names": names.astype("U32")}
    with tempfile.TemporaryDirectory() as td:
        pix = os.path.join(td, "pixel_mat_data")
        os.mkdir(pix)
        for fov in fovs:
            n = 600
            df = pd.DataFrame({"chan0": rs.rand(n)})
            df["fov"] = fov
            lab = rs.randint(0, 16, size=n)
            code = rs.randint(0, 6, size=n)
            code[lab == 0] = 6                                   # one name occurs on background only
            df["label"] = lab
            df["pixel_som_cluster"] = rs.randint(1, 5, size=n)
            df["pixel_meta_cluster_rename"] = names[code]
            feather.write_dataframe(df, os.path.join(pix, fov + ".feather"))
            arrays["lab_" + fov], arrays["code_" + fov] = lab, code
        rows = [(fov, lab, int(rs.randint(20, 200))) for lab in range(1, 16) for fov in fovs]
        cell = pd.DataFrame(rows, columns=["fov", "label", "cell_size"])
        cell_path = os.path.join(td, "cell_table.csv")
        cell.to_csv(cell_path, index=False)
        arrays["cell_fov"] = cell["fov"].values.astype("U8")
        arrays["cell_label"] = cell["label"].values.astype(np.int64)
        arrays["cell_size"] = cell["cell_size"].values.astype(np.int64)
        with warnings.catch_warnings(record=True) as wl:
            warnings.simplefilter("always")
            counts, normed = cell_cluster_utils.create_c2pc_data(fovs, pix, cell_path)
        arrays["warnings"] = np.array([str(w.message) for w in wl if "Pixel clusters" in str(w.message)], dtype="U300")
        for tag, frame in (("counts", counts), ("normed", normed)):
            arrays[f"{tag}_columns"] = np.array(list(frame.columns), dtype="U64")
            arrays[f"{tag}_fov"] = frame["fov"].values.astype("U8")
            arrays[f"{tag}_values"] = frame.drop(columns="fov").values.astype(np.float64)
    save("g8s_c2pc_named", **arrays)


def g9_create_pixel_matrix():
    """The reference's own create_pixel_matrix on a small float32 TIFF cohort (with segmentation masks),
    through the alpineer / skimage stand-ins of tests/golden/_shims (Pillow readers): pre-row-norm channel
    values, pixel threshold, per-FOV tables (full + seeded subset), post-row-norm 99.9 % values, stdout."""
    from PIL import Image
    rs = np.random.RandomState(31)
    fovs, chans = ["fov0", "fov1", "fov2"], ["chan0", "chan1", "chan2", "chan10"]
    arrays = {}
    with tempfile.TemporaryDirectory() as td:
        tiff_dir, seg_dir = os.path.join(td, "tiffs"), os.path.join(td, "seg")
        os.makedirs(os.path.join(td, "pixel_output_dir"))
        os.mkdir(seg_dir)
        for fov in fovs:
            os.makedirs(os.path.join(tiff_dir, fov, "TIFs"))
            for ch in chans:
                img = rs.gamma(0.5, 2.0, size=(28, 24)).astype(np.float32)
                img[rs.uniform(size=img.shape) < 0.4] = 0
                Image.fromarray(img).save(os.path.join(tiff_dir, fov, "TIFs", ch + ".tiff"), format="TIFF")
                arrays[f"img_{fov}_{ch}"] = img
            seg = rs.randint(0, 9, size=(28, 24)).astype(np.int32)
            Image.fromarray(seg).save(os.path.join(seg_dir, fov + "_whole_cell.tiff"), format="TIFF")
            arrays["seg_" + fov] = seg
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            pixie_preprocessing.create_pixel_matrix(list(fovs), list(chans), td, tiff_dir, seg_dir,
                                                    subset_proportion=0.25, seed=42)
        arrays["stdout"] = np.array(buf.getvalue())
        pre = feather.read_dataframe(os.path.join(td, "pixel_output_dir", "channel_norm_pre_rownorm.feather"))
        arrays["pre_columns"] = np.array(list(pre.columns), dtype="U16")
        arrays["pre_values"] = pre.values[0]
        th = feather.read_dataframe(os.path.join(td, "pixel_output_dir", "pixel_thresh.feather"))
        arrays["thresh"] = th["pixel_thresh_val"].values
        post = feather.read_dataframe(os.path.join(td, "channel_norm_post_rownorm.feather"))
        arrays["post_columns"] = np.array(list(post.columns), dtype="U16")
        arrays["post_values"] = post.values[0]
        arrays["data_dir_listing"] = np.array(sorted(os.listdir(os.path.join(td, "pixel_mat_data"))), dtype="U32")
        for fov in fovs:
            for kind in ("pixel_mat_data", "pixel_mat_subsetted"):
                t = feather.read_dataframe(os.path.join(td, kind, fov + ".feather"))
                tag = f"{kind}_{fov}"
                arrays[tag + "_columns"] = np.array(list(t.columns), dtype="U16")
                arrays[tag + "_dtypes"] = np.array([str(d) for d in t.dtypes], dtype="U16")
                arrays[tag + "_channels"] = t[sorted(chans, key=lambda c: int(c[4:]))].values
                arrays[tag + "_meta"] = t[["row_index", "column_index", "label"]].values.astype(np.int64)
    save("g9_create_pixel_matrix", **arrays)


def g10_pixel_cluster_mask():
    """The reference's generate_pixel_cluster_mask (utils/data_utils.py:476-555) on one small FOV table: SOM
    and meta cluster columns (the meta column float-typed, as after a CSV round trip), a mapping table with
    repeated rows and cluster ids that are neither dense nor monotone, pixels missing from the table (mask
    stays 0 there)."""
    from ark.utils import data_utils
    from PIL import Image
    rs = np.random.RandomState(10)
    h, w, k = 37, 41, 12
    keep = np.sort(rs.choice(h * w, size=1100, replace=False))
    som = rs.randint(1, k + 1, size=keep.size)
    som_to_meta = rs.randint(1, 5, size=k + 1)
    meta_to_id = {1: 7, 2: 300, 3: 2, 4: 41}
    table = pd.DataFrame({"chan0": rs.rand(keep.size), "fov": "fov0", "row_index": keep // w, "column_index": keep % w,
                          "pixel_som_cluster": som, "pixel_meta_cluster": som_to_meta[som].astype(np.float64)})
    mapping = pd.DataFrame({"pixel_som_cluster": np.arange(1, k + 1), "pixel_meta_cluster": som_to_meta[1:]})
    mapping["cluster_id"] = [meta_to_id[m] for m in mapping["pixel_meta_cluster"]]
    mapping = pd.concat([mapping, mapping.iloc[:4]], ignore_index=True)           # repeated rows
    som_mapping = mapping.copy()
    som_mapping["cluster_id"] = [(5 * c) % 23 + 1 for c in som_mapping["pixel_som_cluster"]]
    out = {}
    with tempfile.TemporaryDirectory() as td:
        os.makedirs(os.path.join(td, "tiffs", "fov0"))
        os.makedirs(os.path.join(td, "pixel_mat_data"))
        Image.fromarray(rs.rand(h, w).astype(np.float32)).save(os.path.join(td, "tiffs", "fov0", "chan0.tiff"))
        feather.write_dataframe(table, os.path.join(td, "pixel_mat_data", "fov0.feather"))
        for col, mp in (("pixel_meta_cluster", mapping), ("pixel_som_cluster", som_mapping)):
            mask = data_utils.generate_pixel_cluster_mask("fov0", td, os.path.join(td, "tiffs"),
                                                          os.path.join("fov0", "chan0.tiff"), "pixel_mat_data",
                                                          mp, pixel_cluster_col=col)
            out["mask_" + col] = mask
            out["mapping_" + col] = mp[[col, "cluster_id"]].values.astype(np.int64)
    save("g10_pixel_cluster_mask", shape=np.array([h, w]), row_index=table["row_index"].values,
         column_index=table["column_index"].values, pixel_som_cluster=table["pixel_som_cluster"].values,
         pixel_meta_cluster=table["pixel_meta_cluster"].values, **out)


def g14_saved_pixel_masks():
    """The reference's generate_and_save_pixel_cluster_masks (utils/data_utils.py:558-635) on two small FOVs: the
    rewritten cluster-name table and the saved masks."""
    from ark.utils import data_utils
    from PIL import Image
    import tqdm
    data_utils.tqdm = tqdm.tqdm                                  # the notebook bar needs ipywidgets (absent here)
    rs = np.random.RandomState(14)
    h, w, k = 29, 33, 10
    som_to_meta = rs.choice([2, 5, 7, 11], size=k + 1)              # ids 1..4 differ from the labels
    names = pd.DataFrame({"pixel_som_cluster": np.arange(1, k + 1), "pixel_meta_cluster": som_to_meta[1:]})
    names["pixel_meta_cluster_rename"] = ["type_%d" % m for m in names["pixel_meta_cluster"]]
    names["cluster_id"] = 99                                     # a stale column the function must replace
    out = {"shape": np.array([h, w]), "names_text": np.array(names.to_csv(index=False))}
    with tempfile.TemporaryDirectory() as td:
        os.makedirs(os.path.join(td, "pixel_mat_data"))
        os.makedirs(os.path.join(td, "masks"))
        names.to_csv(os.path.join(td, "names.csv"), index=False)
        for fov in ("fov0", "fov1"):
            os.makedirs(os.path.join(td, "tiffs", fov))
            Image.fromarray(rs.rand(h, w).astype(np.float32)).save(os.path.join(td, "tiffs", fov, "chan0.tiff"))
            keep = np.sort(rs.choice(h * w, size=700, replace=False))
            som = rs.randint(1, k + 1, size=keep.size)
            table = pd.DataFrame({"chan0": rs.rand(keep.size), "fov": fov, "row_index": keep // w, "column_index": keep % w,
                                  "pixel_som_cluster": som, "pixel_meta_cluster": som_to_meta[som]})
            feather.write_dataframe(table, os.path.join(td, "pixel_mat_data", fov + ".feather"))
            out["row_index_" + fov], out["column_index_" + fov] = table["row_index"].values, table["column_index"].values
            out["som_" + fov], out["meta_" + fov] = table["pixel_som_cluster"].values, table["pixel_meta_cluster"].values
        data_utils.generate_and_save_pixel_cluster_masks(["fov0", "fov1"], td, os.path.join(td, "masks"), os.path.join(td, "tiffs"),
                                                         "chan0.tiff", "pixel_mat_data", os.path.join(td, "names.csv"),
                                                         pixel_cluster_col="pixel_meta_cluster", sub_dir="pixel_masks",
                                                         name_suffix="_pixel_mask")
        out["names_after_text"] = np.array(open(os.path.join(td, "names.csv")).read())
        for fov in ("fov0", "fov1"):
            with Image.open(os.path.join(td, "masks", "pixel_masks", fov + "_pixel_mask.tiff")) as im:
                out["mask_" + fov] = np.array(im)
    save("g14_saved_pixel_masks", **out)


def g12_cell_meta_clustering():
    """The reference's cell_meta_clustering + add_consensus_labels_cell_table on a seeded cell table."""
    from ark.phenotyping import cell_cluster_utils, cell_meta_clustering
    rs = np.random.RandomState(31)
    cols = ["pixel_meta_cluster_rename_%s" % c for c in ("a", "b", "c", "d", "e", "f")]
    n, k = 700, 30
    data = pd.DataFrame(rs.poisson(3.0, size=(n, 6)) / rs.uniform(50, 500, size=(n, 1)), columns=cols)
    data.insert(0, "cell_size", rs.randint(50, 500, size=n))
    data.insert(1, "fov", rs.choice(["fov0", "fov1", "fov2"], size=n))
    data.insert(2, "segmentation_label", np.arange(1, n + 1))
    data["cell_som_cluster"] = rs.randint(1, k + 1, size=n)
    out = {"cell_values": data[cols].values, "cell_size": data["cell_size"].values.astype(np.int64),
           "fov": data["fov"].values.astype("U8"), "segmentation_label": data["segmentation_label"].values.astype(np.int64),
           "cell_som_cluster": data["cell_som_cluster"].values.astype(np.int64), "cols": np.array(cols)}
    with tempfile.TemporaryDirectory() as td:
        som_avg = cell_cluster_utils.compute_cell_som_cluster_cols_avg(data, cols, "cell_som_cluster", keep_count=True)
        som_avg.to_csv(os.path.join(td, "som_avg.csv"), index=False)
        out["som_avg_text"] = np.array(open(os.path.join(td, "som_avg.csv")).read())
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            cc, assigned = cell_meta_clustering.cell_consensus_cluster(td, cols, data.copy(), "som_avg.csv", max_k=6, cap=3, seed=42)
            cell_meta_clustering.generate_meta_avg_files(td, cc, cols, assigned, "som_avg.csv", "meta_avg.csv")
        out["stdout_consensus"] = np.array(buf.getvalue())
        out["mapping"] = cc.mapping.values.astype(np.int64)
        out["meta_labels"] = assigned["cell_meta_cluster"].values.astype(np.int64)
        out["assigned_columns"] = np.array(list(assigned.columns))
        out["meta_avg_text"] = np.array(open(os.path.join(td, "meta_avg.csv")).read())
        out["som_avg_after_text"] = np.array(open(os.path.join(td, "som_avg.csv")).read())
        # the user's remapping: meta clusters 5 and 6 merged into 5, everything named
        remap = cc.mapping.copy()
        remap["cell_meta_cluster"] = remap["cell_meta_cluster"].replace({6: 5})
        remap["cell_meta_cluster_rename"] = remap["cell_meta_cluster"].map(lambda m: "type_%d" % m)
        remap.to_csv(os.path.join(td, "remap.csv"), index=False)
        out["remap_text"] = np.array(open(os.path.join(td, "remap.csv")).read())
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            remapped = cell_meta_clustering.apply_cell_meta_cluster_remapping(td, assigned.copy(), "remap.csv")
            cell_meta_clustering.generate_remap_avg_count_files(td, remapped, "remap.csv", cols, "som_avg.csv", "meta_avg.csv")
        out["stdout_remap"] = np.array(buf.getvalue())
        out["remapped_meta"] = remapped["cell_meta_cluster"].values.astype(np.int64)
        out["remapped_names"] = remapped["cell_meta_cluster_rename"].values.astype("U16")
        out["meta_avg_remap_text"] = np.array(open(os.path.join(td, "meta_avg.csv")).read())
        out["som_avg_remap_text"] = np.array(open(os.path.join(td, "som_avg.csv")).read())
        # the cell table: every clustered cell plus 25 cells the clustering never saw
        table = pd.DataFrame({"cell_size": np.concatenate([data["cell_size"].values, rs.randint(5, 20, size=25)]),
                              "fov": np.concatenate([data["fov"].values, np.array(["fov1"] * 25)]),
                              "label": np.concatenate([data["segmentation_label"].values, np.arange(n + 1, n + 26)]),
                              "marker": rs.rand(n + 25)})
        table.to_csv(os.path.join(td, "cell_table.csv"), index=False)
        out["cell_table_text"] = np.array(open(os.path.join(td, "cell_table.csv")).read())
        cell_cluster_utils.add_consensus_labels_cell_table(td, os.path.join(td, "cell_table.csv"), remapped.copy())
        out["cell_table_labelled_text"] = np.array(open(os.path.join(td, "cell_table_cell_labels.csv")).read())
    save("g12_cell_meta_clustering", **out)


def g13_weighted_channel():
    """The reference's weighted_channel_comp (its plotting import stubbed out: ark.analysis.visualize is only used by
    the heat-map function, which is not exercised) on a seeded cell-count table."""
    import types
    pkg, vis = types.ModuleType("ark.analysis"), types.ModuleType("ark.analysis.visualize")
    pkg.visualize = vis
    sys.modules.setdefault("ark.analysis", pkg)
    sys.modules.setdefault("ark.analysis.visualize", vis)
    from ark.phenotyping import weighted_channel_comp as wcc
    rs = np.random.RandomState(47)
    chans = ["chan%d" % i for i in range(5)]
    names = ["B", "CD4_T", "CD8_T", "myeloid", "stroma", "tumor", "unassigned"]
    n = 500
    counts = pd.DataFrame(rs.poisson(4.0, size=(n, len(names))).astype(np.float64),
                          columns=["pixel_meta_cluster_rename_%s" % s_ for s_ in names])
    counts.insert(0, "cell_size", rs.randint(50, 400, size=n))
    counts.insert(1, "fov", rs.choice(["fov0", "fov1", "fov2"], size=n))
    counts.insert(2, "segmentation_label", np.arange(1, n + 1))
    pix_avg = pd.DataFrame(rs.rand(len(names), len(chans)), columns=chans)
    pix_avg.insert(0, "pixel_meta_cluster_rename", rs.permutation(names))
    clusters = pd.DataFrame({"fov": counts["fov"].values, "label": counts["segmentation_label"].values,
                             "cell_som_cluster": rs.randint(1, 21, size=n)})
    mapping = pd.DataFrame({"cell_som_cluster": np.arange(1, 21), "cell_meta_cluster": rs.randint(1, 6, size=20)})
    clusters["cell_meta_cluster"] = clusters["cell_som_cluster"].map(dict(mapping.values))
    out = {"chans": np.array(chans), "names": np.array(names), "counts": counts.iloc[:, 3:].values,
           "cell_size": counts["cell_size"].values.astype(np.int64), "fov": counts["fov"].values.astype("U8"),
           "label": counts["segmentation_label"].values.astype(np.int64), "pix_avg": pix_avg[chans].values,
           "pix_avg_ids": pix_avg["pixel_meta_cluster_rename"].values.astype("U16"),
           "cell_som_cluster": clusters["cell_som_cluster"].values.astype(np.int64), "mapping": mapping.values.astype(np.int64)}
    weighted = wcc.compute_p2c_weighted_channel_avg(pix_avg.copy(), chans, counts.copy())
    out["weighted"] = weighted[chans].values
    out["weighted_columns"] = np.array(list(weighted.columns))
    sub = wcc.compute_p2c_weighted_channel_avg(pix_avg.copy(), chans[:3], counts.copy(), fovs=["fov1", "fov2"])
    out["weighted_sub"] = sub[chans[:3]].values
    out["weighted_sub_label"] = sub["label"].values.astype(np.int64)
    with tempfile.TemporaryDirectory() as td:
        feather.write_dataframe(weighted, os.path.join(td, "weighted_cell_channel.feather"), compression="uncompressed")
        cc = types.SimpleNamespace(mapping=mapping)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            wcc.generate_wc_avg_files(["fov0", "fov1", "fov2"], chans, td, cc, clusters.copy())
        out["stdout_wc"] = np.array(buf.getvalue())
        out["som_wc_text"] = np.array(open(os.path.join(td, "cell_som_cluster_channel_avg.csv")).read())
        out["meta_wc_text"] = np.array(open(os.path.join(td, "cell_meta_cluster_channel_avg.csv")).read())
        remap = mapping.copy()
        remap["cell_meta_cluster"] = remap["cell_meta_cluster"].replace({5: 4})
        remap["cell_meta_cluster_rename"] = remap["cell_meta_cluster"].map(lambda m_: "type_%d" % m_)
        remap.to_csv(os.path.join(td, "remap.csv"), index=False)
        out["remap_text"] = np.array(open(os.path.join(td, "remap.csv")).read())
        remapped = clusters.copy()
        remapped["cell_meta_cluster"] = remapped["cell_som_cluster"].map(dict(remap[["cell_som_cluster", "cell_meta_cluster"]].values))
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            wcc.generate_remap_avg_wc_files(["fov0", "fov1", "fov2"], chans, td, remapped, "remap.csv",
                                            "weighted_cell_channel.feather", "cell_som_cluster_channel_avg.csv",
                                            "cell_meta_cluster_channel_avg.csv")
        out["stdout_remap"] = np.array(buf.getvalue())
        out["som_wc_remap_text"] = np.array(open(os.path.join(td, "cell_som_cluster_channel_avg.csv")).read())
        out["meta_wc_remap_text"] = np.array(open(os.path.join(td, "cell_meta_cluster_channel_avg.csv")).read())
    save("g13_weighted_channel", **out)


if __name__ == "__main__":
    ob.build()
    steps = {"g1": g1_normalize, "g2": g2_g5_preprocess, "g3": g3_quantiles, "g4": g4_cluster_avg, "g5": g5_meta_clustering, "g6": g6_som, "g7b": g7b_batch_mode,
             "g7": g7_end_to_end, "g8": g8_c2pc, "g8s": g8s_c2pc_named, "g9": g9_create_pixel_matrix, "g10": g10_pixel_cluster_mask, "g12": g12_cell_meta_clustering, "g13": g13_weighted_channel, "g14": g14_saved_pixel_masks}
    for name in (sys.argv[1:] or list(steps)):
        steps[name]()
