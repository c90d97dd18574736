"""The pre-processing kernels of one FOV (create_fov_pixel_data + the 99.9 % values; reference
pixie_preprocessing.py:47-75, 406-408) on one synthetic 1024^2 x 22 binary64 image in HBM, several repetitions --
the command rocprofv3 wraps for profiles/rNN/preprocess.txt (kernel trace, FETCH_SIZE / WRITE_SIZE passes)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ark_analysis_amd import som_device  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--side", type=int, default=1024)
ap.add_argument("--channels", type=int, default=22)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--f32", action="store_true", help="float32 semantics (the pipeline's TIFF dtype)")
a = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(0)
h = w = a.side
c = a.channels
img0 = torch.empty((h, w, c), dtype=torch.float64, device=dev).exponential_(1.0, generator=g)
img0.mul_((torch.rand((h, w, c), generator=g, device=dev) >= 0.4).to(torch.float64))
if a.f32:
    img0 = img0.float().double()
tmp = torch.empty_like(img0)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
times = {"blur": [], "rowfilter": [], "quantile": []}
for r in range(a.reps + 1):
    img = img0.clone()
    ev[0].record()
    som_device.gaussian_blur_hwc(img, 2.0, tmp=tmp, f32_semantics=a.f32)
    ev[1].record()
    rows, index = som_device.rowsum_filter_normalize(img.view(h * w, c), 0.0, f32_semantics=a.f32)
    ev[2].record()
    q = som_device.quantile_nonzero(rows, 0.999)
    ev[3].record()
    torch.cuda.synchronize()
    if r:
        times["blur"].append(ev[0].elapsed_time(ev[1]))
        times["rowfilter"].append(ev[1].elapsed_time(ev[2]))
        times["quantile"].append(ev[2].elapsed_time(ev[3]))
px = h * w
print(json.dumps({"workload": f"{h}x{w}x{c} binary64 image", "kept_rows": int(rows.shape[0]),
                  **{k + "_ms": round(sum(v) / len(v), 4) for k, v in times.items()},
                  "blur_GBps_algorithmic": round(4 * px * c * 8 / (sum(times["blur"]) / len(times["blur"]) * 1e-3) / 1e9, 1)}))
