"""The drop-in pipeline under a 2-rank process group on CPU (gloo): train_pixel_som (batch mode: training tables
sharded by rank, statistics all-reduced; online mode: rank 0 trains, codebook broadcast) -> cluster_pixels (FOV
files dealt round robin, som_clusters_seen united, one rank swaps the directories) -> generate_som_avg_files
(totals all-reduced, rank 0 writes).  The device entry points are the oracle stand-ins of tests/oracle_backend.py
(test infrastructure): this exercises the host logic; the same functions run on the HIP path in the -m gpu tests.
Reference loops being sharded: /root/reference/src/ark/phenotyping/pixel_som_clustering.py:250-285,
pixel_cluster_utils.py:369-404."""
import os
import socket

import numpy as np
import pandas as pd
import pytest
import torch.multiprocessing as mp

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CHANS = ["chan%d" % i for i in range(4)]
FOVS = ["fov0", "fov1", "fov2"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, td, mode, out_path, fovs=None, kernels="oracle"):
    fovs = list(FOVS) if fovs is None else list(fovs)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import io
    import contextlib
    import torch.distributed as dist
    from ark_analysis_amd import flowsom  # noqa: F401
    from ark_analysis_amd.phenotyping import pixel_som_clustering
    from ark_analysis_amd.phenotyping.cluster_helpers import read_dataframe
    if kernels == "oracle":
        from tests import oracle_backend
        oracle_backend.install(setattr)
    else:
        # the HIP path on ONE device shared by the ranks: RCCL refuses two ranks on a GPU, so the group is gloo
        # (joined here, before the pipeline's own init_from_env would pick "nccl")
        import torch
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        if kernels.startswith("hip+"):     # the in-library exchange over a stand-in collective library
            os.environ.update(PXSOM_RCCL_LIBRARY=kernels[4:], PXSOM_NATIVE_EXCHANGE="force")
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        obj = pixel_som_clustering.train_pixel_som(fovs, CHANS, td, num_passes=1, seed=42, train_mode=mode,
                                                   batch_steps=4)
        n_train = len(obj.train_data)
        pixel_som_clustering.cluster_pixels(fovs, td, obj)
        pixel_som_clustering.generate_som_avg_files(fovs, CHANS, td, obj, data_dir="pixel_mat_data")
    assert dist.get_world_size() == world
    from ark_analysis_amd import distributed as _d
    in_library = any(c is not None for c in _d._native_comms.values())
    np.savez(out_path % rank, weights=obj.weights.values, n_train=n_train, stdout=np.array(buf.getvalue()),
             in_library=in_library,
             seen=np.array(sorted(int(v) for v in obj.som_clusters_seen), dtype=np.int64),
             file_weights=read_dataframe(os.path.join(td, "pixel_som_weights.feather")).values)
    dist.barrier()
    dist.destroy_process_group()


def _build(td, g):
    from ark_analysis_amd.phenotyping.cluster_helpers import write_dataframe
    os.mkdir(os.path.join(td, "pixel_mat_data"))
    os.mkdir(os.path.join(td, "pixel_mat_subsetted"))
    write_dataframe(pd.DataFrame(g["norm"][None, :], columns=CHANS), os.path.join(td, "post_rowsum_chan_norm.feather"))
    for fov in FOVS:
        df = pd.DataFrame(g["data_" + fov], columns=CHANS)
        df["fov"] = fov
        meta = g["meta_" + fov]
        df["row_index"], df["column_index"], df["label"] = meta[:, 0], meta[:, 1], meta[:, 2]
        write_dataframe(df, os.path.join(td, "pixel_mat_data", fov + ".feather"))
        write_dataframe(df.iloc[g["subidx_" + fov]], os.path.join(td, "pixel_mat_subsetted", fov + ".feather"))


@pytest.mark.parametrize("mode", ["batch", "online"])
def test_two_rank_pipeline(oracle, tmp_path, mode):
    from ark_analysis_amd.flowsom import default_radius_range
    from ark_analysis_amd.phenotyping.cluster_helpers import read_dataframe
    g = np.load(os.path.join(GOLD, "g7_pixel_pipeline.npz"))
    td = str(tmp_path / "job")
    os.mkdir(td)
    _build(td, g)
    out = str(tmp_path / "rank%d.npz")
    mp.spawn(_worker, args=(2, _free_port(), td, mode, out), nprocs=2, join=True)
    r0, r1 = np.load(out % 0), np.load(out % 1)
    # every rank ends with the same codebook; rank 0 wrote it
    np.testing.assert_array_equal(r0["weights"], r1["weights"])
    np.testing.assert_array_equal(r0["file_weights"], r0["weights"])
    norm = g["norm"]
    sub = {fov: g["sub_" + fov] / norm for fov in FOVS}
    m = 4
    if mode == "batch":
        # FOVs are dealt round robin: rank 0 trains on fov0 + fov2, rank 1 on fov1
        assert int(r0["n_train"]) == 800 and int(r1["n_train"]) == 400
        local = [np.concatenate([sub["fov0"], sub["fov2"]]), sub["fov1"]]
        init = local[0][np.random.RandomState(42).choice(len(local[0]), 100, replace=False)]
        # single-process equivalent: blocks of m rows interleaved, so that global row i % m selects the union of
        # the ranks' local mini-batches (i % m)
        blocks = []
        for j in range(len(local[0]) // m):
            blocks.append(local[0][j * m:(j + 1) * m])
            if (j + 1) * m <= len(local[1]):
                blocks.append(local[1][j * m:(j + 1) * m])
        want = oracle.som_batch(np.concatenate(blocks), init, 10, 10, 1, (0.05, 0.01), default_radius_range(10, 10), m)
        np.testing.assert_allclose(r0["weights"], want, rtol=1e-10, atol=0)
    else:
        # the online rule is sequential: rank 0 trains on all three tables, exactly as a single process does
        assert int(r0["n_train"]) == int(r1["n_train"]) == 1200
        np.testing.assert_array_equal(r0["weights"], g["weights"])
    # labels of every table == the oracle's for the trained codebook; tables normalised; staging directory gone
    w = r0["weights"]
    counts = np.zeros(100, dtype=np.int64)
    sums = np.zeros((100, 4))
    for fov in FOVS:
        res = read_dataframe(os.path.join(td, "pixel_mat_data", fov + ".feather"))
        np.testing.assert_array_equal(res[CHANS].values, g["normed_" + fov])
        want_l, _ = oracle.map_data_to_nodes(w, res[CHANS].values)
        np.testing.assert_array_equal(res["pixel_som_cluster"].values, want_l)
        s, c = oracle.cluster_sums(res[CHANS].values, want_l, 100)
        sums += s
        counts += c
    assert not os.path.exists(os.path.join(td, "pixel_mat_data_temp"))
    seen = np.flatnonzero(counts) + 1
    np.testing.assert_array_equal(r0["seen"], seen)      # united over the ranks, on every rank
    np.testing.assert_array_equal(r1["seen"], seen)
    avg = pd.read_csv(os.path.join(td, "pixel_channel_avg_som_cluster.csv"))
    np.testing.assert_array_equal(avg["pixel_som_cluster"].values, seen)
    np.testing.assert_array_equal(avg["count"].values, counts[seen - 1])
    np.testing.assert_allclose(avg[CHANS].values, sums[seen - 1] / counts[seen - 1][:, None], rtol=1e-12, atol=0)
    # rank 0 speaks for the job, the other ranks stay silent
    assert str(r0["stdout"]) == ("Training SOM\nMapping pixel data to SOM cluster labels\nProcessed 3 fovs\n"
                                 "Computing average channel expression across pixel SOM clusters\n")
    assert str(r1["stdout"]) == ""
    if mode == "online":
        for fov in FOVS:
            res = read_dataframe(os.path.join(td, "pixel_mat_data", fov + ".feather"))
            np.testing.assert_array_equal(res["pixel_som_cluster"].values, g["labels_" + fov])


def test_more_ranks_than_fovs(oracle, tmp_path):
    """Three ranks, two FOV tables: the rank that is dealt nothing trains on no rows (and still takes part in every
    exchange), labels no table, and ends with the same codebook and cluster set as the others."""
    from ark_analysis_amd.phenotyping.cluster_helpers import read_dataframe
    g = np.load(os.path.join(GOLD, "g7_pixel_pipeline.npz"))
    td = str(tmp_path / "job")
    os.mkdir(td)
    _build(td, g)
    for fov in FOVS[2:]:
        for folder in ("pixel_mat_data", "pixel_mat_subsetted"):
            os.remove(os.path.join(td, folder, fov + ".feather"))
    out = str(tmp_path / "rank%d.npz")
    mp.spawn(_worker, args=(3, _free_port(), td, "batch", out, FOVS[:2]), nprocs=3, join=True)
    ranks = [np.load(out % r) for r in range(3)]
    assert [int(r["n_train"]) for r in ranks] == [400, 400, 0]
    for r in ranks[1:]:
        np.testing.assert_array_equal(r["weights"], ranks[0]["weights"])
        np.testing.assert_array_equal(r["seen"], ranks[0]["seen"])
    w = ranks[0]["weights"]
    for fov in FOVS[:2]:
        res = read_dataframe(os.path.join(td, "pixel_mat_data", fov + ".feather"))
        want_l, _ = oracle.map_data_to_nodes(w, res[CHANS].values)
        np.testing.assert_array_equal(res["pixel_som_cluster"].values, want_l)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["batch", "online", "batch, exchange inside the library"])
def test_two_rank_pipeline_on_the_hip_path(oracle, tmp_path, mode):
    """The same three pipeline functions on two ranks with the real kernels (both ranks on device 0, gloo group):
    Arrow labelling path, recycled host blocks, totals cache, rank-sharded files -- against the oracle.  Third
    variant: train_pixel_som -> BatchSOMTrainer -> distributed.native_exchange -> pxsom_batch_train_steps_sharded, the
    collective library replaced by tests/mock_rccl (RCCL does not put two ranks on one GPU)."""
    kernels = "hip"
    if mode.endswith("library"):
        from tests.test_gpu_exchange import _mock_library
        kernels, mode = "hip+" + _mock_library(tmp_path), "batch"
    from ark_analysis_amd.flowsom import default_radius_range
    from ark_analysis_amd.phenotyping.cluster_helpers import read_dataframe
    g = np.load(os.path.join(GOLD, "g7_pixel_pipeline.npz"))
    td = str(tmp_path / "job")
    os.mkdir(td)
    _build(td, g)
    out = str(tmp_path / "rank%d.npz")
    mp.spawn(_worker, args=(2, _free_port(), td, mode, out, None, kernels), nprocs=2, join=True)
    r0, r1 = np.load(out % 0), np.load(out % 1)
    np.testing.assert_array_equal(r0["weights"], r1["weights"])
    assert bool(r0["in_library"]) == bool(r1["in_library"]) == kernels.startswith("hip+")   # (no silent fallback)
    w = r0["weights"]
    if mode == "online":
        np.testing.assert_array_equal(w, g["weights"])               # the reference-order run: bit-equal
    else:
        norm = g["norm"]
        sub = {fov: g["sub_" + fov] / norm for fov in FOVS}
        local = [np.concatenate([sub["fov0"], sub["fov2"]]), sub["fov1"]]
        init = local[0][np.random.RandomState(42).choice(len(local[0]), 100, replace=False)]
        blocks, m = [], 4
        for j in range(len(local[0]) // m):
            blocks.append(local[0][j * m:(j + 1) * m])
            if (j + 1) * m <= len(local[1]):
                blocks.append(local[1][j * m:(j + 1) * m])
        want = oracle.som_batch(np.concatenate(blocks), init, 10, 10, 1, (0.05, 0.01), default_radius_range(10, 10), m)
        np.testing.assert_allclose(w, want, rtol=1e-9, atol=0)
    counts = np.zeros(100, dtype=np.int64)
    sums = np.zeros((100, 4))
    for fov in FOVS:
        res = read_dataframe(os.path.join(td, "pixel_mat_data", fov + ".feather"))
        np.testing.assert_array_equal(res[CHANS].values, g["normed_" + fov])
        want_l, _ = oracle.map_data_to_nodes(w, res[CHANS].values)
        np.testing.assert_array_equal(res["pixel_som_cluster"].values, want_l)
        s_, c_ = oracle.cluster_sums(res[CHANS].values, want_l, 100)
        sums += s_
        counts += c_
    seen = np.flatnonzero(counts) + 1
    np.testing.assert_array_equal(r0["seen"], seen)
    np.testing.assert_array_equal(r1["seen"], seen)
    avg = pd.read_csv(os.path.join(td, "pixel_channel_avg_som_cluster.csv"))
    np.testing.assert_array_equal(avg["count"].values, counts[seen - 1])
    np.testing.assert_allclose(avg[CHANS].values, sums[seen - 1] / counts[seen - 1][:, None], rtol=1e-12, atol=0)


def _matrix_worker(rank, world, port, td, out_path, kernels="oracle"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import io
    import contextlib
    import torch.distributed as dist
    from ark_analysis_amd.phenotyping import pixie_preprocessing
    if kernels == "oracle":
        from tests import oracle_backend
        oracle_backend.install(setattr)
    else:     # both ranks on device 0: the group has to be gloo (joined before init_from_env would pick "nccl")
        import torch
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    fovs, chans = ["fov0", "fov1", "fov2"], ["chan0", "chan1", "chan2", "chan10"]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        pixie_preprocessing.create_pixel_matrix(list(fovs), list(chans), td, os.path.join(td, "tiffs"),
                                                os.path.join(td, "seg"), subset_proportion=0.25, seed=42)
        pixie_preprocessing.create_pixel_matrix(list(fovs), list(chans), td, os.path.join(td, "tiffs"),
                                                os.path.join(td, "seg"), subset_proportion=0.25, seed=42)
    assert dist.get_world_size() == world
    with open(out_path % rank, "w") as f:
        f.write(buf.getvalue())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kernels", ["oracle", pytest.param("hip", marks=pytest.mark.gpu)])
def test_create_pixel_matrix_on_two_ranks(oracle, tmp_path, kernels):
    """create_pixel_matrix under a 2-rank group (reference loop being sharded: pixie_preprocessing.py:375-432): the
    FOVs' tables are made by different ranks, the per-FOV values behind the three normalisation files are gathered
    and averaged in one agreed order -- every file equals the reference's single-process run (g9).  On CPU with the
    oracle stand-ins, on the GPU box with the real kernels (both ranks on device 0)."""
    from tests.test_pipeline_dropin import _write_g9_cohort
    from ark_analysis_amd.phenotyping.cluster_helpers import read_dataframe
    g = np.load(os.path.join(GOLD, "g9_create_pixel_matrix.npz"))
    td = str(tmp_path / "job")
    os.mkdir(td)
    fovs, chans, _, _ = _write_g9_cohort(g, td)
    out = str(tmp_path / "rank%d.txt")
    mp.spawn(_matrix_worker, args=(2, _free_port(), td, out, kernels), nprocs=2, join=True)
    # rank 0 reports; the second call finds nothing to do
    assert open(out % 0).read() == "Processed 3 fovs\nThere are no more FOVs to preprocess, skipping\n"
    assert open(out % 1).read() == ""
    pre = read_dataframe(os.path.join(td, "pixel_output_dir", "channel_norm_pre_rownorm.feather"))
    assert list(pre.columns) == list(g["pre_columns"]) and pre.values.dtype == g["pre_values"].dtype
    np.testing.assert_array_equal(pre.values[0], g["pre_values"])
    th = read_dataframe(os.path.join(td, "pixel_output_dir", "pixel_thresh.feather"))["pixel_thresh_val"].values
    np.testing.assert_array_equal(th, g["thresh"])
    post = read_dataframe(os.path.join(td, "channel_norm_post_rownorm.feather"))
    assert list(post.columns) == list(g["post_columns"])
    # (a mean over the per-FOV columns in the order they were recorded: set order in the reference, the agreed
    # to-do order here -- the last place may differ)
    np.testing.assert_allclose(post.values[0], g["post_values"], rtol=1e-15, atol=0)
    assert sorted(os.listdir(os.path.join(td, "pixel_mat_data"))) == list(g["data_dir_listing"])   # no record left
    for fov in fovs:
        for kind in ("pixel_mat_data", "pixel_mat_subsetted"):
            t = read_dataframe(os.path.join(td, kind, fov + ".feather"))
            tag = f"{kind}_{fov}"
            assert list(t.columns) == list(g[tag + "_columns"])
            np.testing.assert_array_equal(t[["chan0", "chan1", "chan2", "chan10"]].values, g[tag + "_channels"])
            np.testing.assert_array_equal(t[["row_index", "column_index", "label"]].values.astype(np.int64),
                                          g[tag + "_meta"])
