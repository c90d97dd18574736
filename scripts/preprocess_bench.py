"""create_pixel_matrix end to end on a synthetic float32 TIFF cohort (SURVEY.md section 8 f, rank 2): N FOVs of
side^2 x C single-channel TIFFs + a segmentation mask each, in a scratch directory.  Reports wall time per
stage (the two percentile passes over the TIFFs, then blur / filter / normalise / write per FOV) and Mpx/s."""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ark_analysis_amd import image_io  # noqa: E402
from ark_analysis_amd.phenotyping import pixel_cluster_utils, pixie_preprocessing  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--fovs", type=int, default=3)
ap.add_argument("--side", type=int, default=1024)
ap.add_argument("--channels", type=int, default=22)
ap.add_argument("--scratch", default=None)
args = ap.parse_args()

root = tempfile.mkdtemp(prefix="pxsom_pre_", dir=args.scratch)
tiff_dir, seg_dir = os.path.join(root, "tiffs"), os.path.join(root, "seg")
os.makedirs(os.path.join(root, "pixel_output_dir"))
os.mkdir(seg_dir)
fovs = ["fov%d" % i for i in range(args.fovs)]
chans = ["chan%d" % i for i in range(args.channels)]
rs = np.random.RandomState(0)
for fov in fovs:
    os.makedirs(os.path.join(tiff_dir, fov, "TIFs"))
    for ch in chans:
        img = rs.gamma(0.5, 2.0, size=(args.side, args.side)).astype(np.float32)
        img[rs.uniform(size=img.shape) < 0.4] = 0
        image_io.write_channel(os.path.join(tiff_dir, fov, "TIFs", ch + ".tiff"), img)
    image_io.write_channel(os.path.join(seg_dir, fov + "_whole_cell.tiff"),
                           rs.randint(0, 2000, size=(args.side, args.side)).astype(np.int32))

from ark_analysis_amd import flowsom  # noqa: E402
flowsom.positive_quantile_f32(np.ones((8, 8), dtype=np.float32), 0.5)   # library load / HIP init outside the timings
t0 = time.perf_counter()
pre = pixel_cluster_utils.calculate_channel_percentiles(tiff_dir, fovs, chans, "TIFs", 0.99)
t1 = time.perf_counter()
thr = pixel_cluster_utils.calculate_pixel_intensity_percentile(tiff_dir, fovs, list(pre.columns), "TIFs", pre)
t2 = time.perf_counter()
pixie_preprocessing.create_pixel_matrix(list(fovs), list(chans), root, tiff_dir, seg_dir)
t3 = time.perf_counter()
px = args.side * args.side * args.fovs
print(json.dumps({"workload": f"{args.fovs} FOVs {args.side}^2 x {args.channels} float32 TIFFs + mask, scratch {root}",
                  "channel_percentiles_s": round(t1 - t0, 3), "pixel_threshold_s": round(t2 - t1, 3),
                  "create_pixel_matrix_total_s": round(t3 - t2, 3),
                  "create_pixel_matrix_Mpx_per_s": round(px / (t3 - t2) / 1e6, 2)}))
shutil.rmtree(root)
