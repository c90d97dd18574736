"""weighted_channel_comp (without its plotting function) against the reference's own run
(tests/golden/g13_weighted_channel.npz, tests/golden/make_golden.py g13): the count-weighted channel expression of
every cell, its per-cluster averages, the files after a manual remapping."""
import io
import os
import types

import numpy as np
import pandas as pd
import pytest

from ark_analysis_amd.fov_tables import write_dataframe
from ark_analysis_amd.phenotyping import weighted_channel_comp as wcc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _frames_equal(text_got, text_want):
    got, want = pd.read_csv(io.StringIO(text_got)), pd.read_csv(io.StringIO(str(text_want)))
    assert list(got.columns) == list(want.columns)
    for col in want.columns:
        if want[col].dtype.kind in "fc":
            np.testing.assert_allclose(got[col].values, want[col].values, rtol=1e-12, atol=0, err_msg=col)
        else:
            assert list(got[col].values) == list(want[col].values), col


def test_weighted_channel_pipeline_matches_reference_run(tmp_path, capsys):
    g = np.load(os.path.join(GOLD, "g13_weighted_channel.npz"))
    td = str(tmp_path)
    chans, names = list(g["chans"]), list(g["names"])
    counts = pd.DataFrame(g["counts"], columns=["pixel_meta_cluster_rename_%s" % s for s in names])
    counts.insert(0, "cell_size", g["cell_size"])
    counts.insert(1, "fov", g["fov"])
    counts.insert(2, "segmentation_label", g["label"])
    pix_avg = pd.DataFrame(g["pix_avg"], columns=chans)
    pix_avg.insert(0, "pixel_meta_cluster_rename", g["pix_avg_ids"])
    mapping = pd.DataFrame(g["mapping"], columns=["cell_som_cluster", "cell_meta_cluster"])
    clusters = pd.DataFrame({"fov": g["fov"], "label": g["label"], "cell_som_cluster": g["cell_som_cluster"]})
    clusters["cell_meta_cluster"] = clusters["cell_som_cluster"].map(dict(mapping.values))

    weighted = wcc.compute_p2c_weighted_channel_avg(pix_avg.copy(), chans, counts.copy())
    assert list(weighted.columns) == list(g["weighted_columns"])
    np.testing.assert_allclose(weighted[chans].values, g["weighted"], rtol=1e-13, atol=0)
    sub = wcc.compute_p2c_weighted_channel_avg(pix_avg.copy(), chans[:3], counts.copy(), fovs=["fov1", "fov2"])
    np.testing.assert_allclose(sub[chans[:3]].values, g["weighted_sub"], rtol=1e-13, atol=0)
    np.testing.assert_array_equal(sub["label"].values, g["weighted_sub_label"])
    with pytest.raises(ValueError):
        wcc.compute_p2c_weighted_channel_avg(pix_avg.copy(), chans, counts.copy(), fovs=["fov9"])
    with pytest.raises(ValueError):
        wcc.compute_p2c_weighted_channel_avg(pix_avg.copy(), chans, counts.copy(), pixel_cluster_col="pixel_meta_cluster")
    with pytest.raises(ValueError):          # a cluster the average table does not hold
        wcc.compute_p2c_weighted_channel_avg(pix_avg.iloc[1:].copy(), chans, counts.copy())

    write_dataframe(weighted, os.path.join(td, "weighted_cell_channel.feather"))
    cc = types.SimpleNamespace(mapping=mapping)
    capsys.readouterr()
    wcc.generate_wc_avg_files(["fov0", "fov1", "fov2"], chans, td, cc, clusters.copy())
    assert capsys.readouterr().out == str(g["stdout_wc"])
    _frames_equal(open(os.path.join(td, "cell_som_cluster_channel_avg.csv")).read(), g["som_wc_text"])
    _frames_equal(open(os.path.join(td, "cell_meta_cluster_channel_avg.csv")).read(), g["meta_wc_text"])
    wcc.generate_wc_avg_files(["fov0", "fov1", "fov2"], chans, td, cc, clusters.copy())
    assert capsys.readouterr().out == "Already generated average weighted channel expression files, skipping\n"

    with open(os.path.join(td, "remap.csv"), "w") as f:
        f.write(str(g["remap_text"]))
    remap = pd.read_csv(os.path.join(td, "remap.csv"))
    remapped = clusters.copy()
    remapped["cell_meta_cluster"] = remapped["cell_som_cluster"].map(dict(remap[["cell_som_cluster", "cell_meta_cluster"]].values))
    wcc.generate_remap_avg_wc_files(["fov0", "fov1", "fov2"], chans, td, remapped, "remap.csv", "weighted_cell_channel.feather",
                                    "cell_som_cluster_channel_avg.csv", "cell_meta_cluster_channel_avg.csv")
    assert capsys.readouterr().out == str(g["stdout_remap"])
    _frames_equal(open(os.path.join(td, "cell_som_cluster_channel_avg.csv")).read(), g["som_wc_remap_text"])
    _frames_equal(open(os.path.join(td, "cell_meta_cluster_channel_avg.csv")).read(), g["meta_wc_remap_text"])
    # cells in another order than the weighted table's are matched by (fov, label); a missing cell is refused
    shuffled = clusters.sample(frac=1.0, random_state=1)
    got = wcc.compute_cell_cluster_weighted_channel_avg(["fov0", "fov1", "fov2"], chans, td, "weighted_cell_channel.feather",
                                                        shuffled, "cell_som_cluster")
    _frames_equal(got.to_csv(index=False), pd.read_csv(io.StringIO(str(g["som_wc_text"])))[["cell_som_cluster"] + chans].to_csv(index=False))
    with pytest.raises(ValueError):
        wcc.compute_cell_cluster_weighted_channel_avg(["fov0", "fov1", "fov2"], chans, td, "weighted_cell_channel.feather",
                                                      clusters.iloc[1:], "cell_som_cluster")
