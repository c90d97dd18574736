"""End-to-end ``cluster_pixels`` on feather tables (SURVEY.md section 8 f, rank 1): what a notebook user
waits for once the BMU search itself is fast.  Writes N synthetic 1024^2 x 22 FOV tables (float64, the
reference's on-disk format) to a scratch directory, trains the pixel SOM on the 10 % subset tables, then
times cluster_pixels and, separately, its three stages (read all, label all, write all) to show how much
of the feather round trip the reader / writer threads hide."""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np
import pandas as pd
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ark_analysis_amd import fov_tables, synth  # noqa: E402
from ark_analysis_amd.phenotyping import pixel_som_clustering  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--fovs", type=int, default=4)
ap.add_argument("--side", type=int, default=1024)
ap.add_argument("--channels", type=int, default=22)
ap.add_argument("--scratch", default=None)
args = ap.parse_args()

root = tempfile.mkdtemp(prefix="pxsom_pipe_", dir=args.scratch)
chans = ["chan%d" % i for i in range(args.channels)]
fovs = ["fov%d" % i for i in range(args.fovs)]
n = args.side * args.side
os.mkdir(os.path.join(root, "pixel_mat_data"))
os.mkdir(os.path.join(root, "pixel_mat_subsetted"))
t0 = time.perf_counter()
for i, fov in enumerate(fovs):
    x = synth.make_fov_numpy(n, args.channels, seed=1000 + i, dtype=np.float64)
    df = pd.DataFrame(x, columns=chans)
    df["fov"] = fov
    df["row_index"] = np.repeat(np.arange(args.side), args.side)
    df["column_index"] = np.tile(np.arange(args.side), args.side)
    df["label"] = 0
    fov_tables.write_dataframe(df, os.path.join(root, "pixel_mat_data", fov + ".feather"))
    fov_tables.write_dataframe(df.iloc[::10], os.path.join(root, "pixel_mat_subsetted", fov + ".feather"))
fov_tables.write_dataframe(pd.DataFrame(np.ones((1, args.channels)), columns=chans),
                           os.path.join(root, "post_rowsum_chan_norm.feather"))
t_gen = time.perf_counter() - t0
bytes_per_table = os.path.getsize(os.path.join(root, "pixel_mat_data", fovs[0] + ".feather"))

t0 = time.perf_counter()
som = pixel_som_clustering.train_pixel_som(fovs, chans, root)
torch.cuda.synchronize()
t_train = time.perf_counter() - t0

tables = fov_tables.FovTableDir(os.path.join(root, "pixel_mat_data"))
# stage timings, one after the other
t0 = time.perf_counter()
loaded = [tables.load(f) for f in fovs]
t_read = time.perf_counter() - t0
t0 = time.perf_counter()
labelled = [som.assign_som_clusters(t) for t in loaded]
torch.cuda.synchronize()
t_label = time.perf_counter() - t0
os.mkdir(os.path.join(root, "w"))
t0 = time.perf_counter()
for f, t in zip(fovs, labelled):
    fov_tables.write_dataframe(t, os.path.join(root, "w", f + ".feather"))
t_write = time.perf_counter() - t0
del loaded, labelled

t0 = time.perf_counter()
pixel_som_clustering.cluster_pixels(fovs, root, som)
t_pipe = time.perf_counter() - t0
px = n * args.fovs
# generate_som_avg_files straight after cluster_pixels: the per-cluster totals were taken while the rows were in HBM
t0 = time.perf_counter()
pixel_som_clustering.generate_som_avg_files(fovs, chans, root, som, data_dir="pixel_mat_data", num_fovs_subset=len(fovs))
t_avg_cached = time.perf_counter() - t0
# the per-SOM-cluster mean table over the labelled tables (generate_som_avg_files' work): device path, then
# the DataFrame path the reference shape implies (same kernel underneath)
from ark_analysis_amd.phenotyping import pixel_cluster_utils  # noqa: E402
chans = list(som.weights.columns)
t0 = time.perf_counter()
avg = pixel_cluster_utils.compute_pixel_cluster_channel_avg(fovs, chans, root, "pixel_som_cluster", None, "pixel_mat_data",
                                                            num_fovs_subset=len(fovs), keep_count=True)
t_avg = time.perf_counter() - t0
device_sums = pixel_cluster_utils._DEVICE_SUMS
pixel_cluster_utils._DEVICE_SUMS = None           # forces the DataFrame route
t0 = time.perf_counter()
avg_df = pixel_cluster_utils.compute_pixel_cluster_channel_avg(fovs, chans, root, "pixel_som_cluster", None,
                                                               "pixel_mat_data", num_fovs_subset=len(fovs), keep_count=True)
t_avg_df = time.perf_counter() - t0
pixel_cluster_utils._DEVICE_SUMS = device_sums
assert np.array_equal(avg["count"].values, avg_df["count"].values)
assert np.allclose(avg[chans].values, avg_df[chans].values, rtol=1e-13, atol=0)
print(json.dumps({
    "workload": f"{args.fovs} FOV tables {args.side}^2 x {args.channels} float64 + 4 meta columns, "
                f"{bytes_per_table / 1e6:.0f} MB each, scratch {root}",
    "train_pixel_som_s": round(t_train, 3),
    "stages_sequential_s": {"read": round(t_read, 3), "normalise+label (incl. PCIe)": round(t_label, 3),
                            "write": round(t_write, 3), "sum": round(t_read + t_label + t_write, 3)},
    "cluster_pixels_s": round(t_pipe, 3),
    "cluster_pixels_Mpx_per_s": round(px / t_pipe / 1e6, 2),
    "generate_som_avg_files_after_cluster_pixels_s": round(t_avg_cached, 3),
    "cluster_channel_avg_s": round(t_avg, 3), "cluster_channel_avg_dataframe_route_s": round(t_avg_df, 3),
    "generated_in_s": round(t_gen, 1)}))
fov_tables.wait_for_cleanup()
shutil.rmtree(root)
