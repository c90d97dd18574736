"""Where create_pixel_matrix's per-FOV time goes (synthetic float32 TIFF cohort as scripts/preprocess_bench.py)."""
import argparse, os, shutil, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ark_analysis_amd import flowsom, fov_tables, image_io
from ark_analysis_amd.phenotyping import pixel_cluster_utils, pixie_preprocessing as pp
ap = argparse.ArgumentParser(); ap.add_argument("--fovs", type=int, default=6); args = ap.parse_args()
if "PXSOM_SWITCH" in os.environ: sys.setswitchinterval(float(os.environ["PXSOM_SWITCH"]))   # GIL hand-over interval (default 5 ms)
root = tempfile.mkdtemp(prefix="pxsom_pre_")
tiff_dir, seg_dir = os.path.join(root, "tiffs"), os.path.join(root, "seg")
os.makedirs(os.path.join(root, "pixel_output_dir")); os.mkdir(seg_dir)
fovs = ["fov%d" % i for i in range(args.fovs)]; chans = ["chan%d" % i for i in range(22)]
rs = np.random.RandomState(0)
for fov in fovs:
    os.makedirs(os.path.join(tiff_dir, fov, "TIFs"))
    for ch in chans:
        img = rs.gamma(0.5, 2.0, size=(1024, 1024)).astype(np.float32); img[rs.uniform(size=img.shape) < 0.4] = 0
        image_io.write_channel(os.path.join(tiff_dir, fov, "TIFs", ch + ".tiff"), img)
    image_io.write_channel(os.path.join(seg_dir, fov + "_whole_cell.tiff"), rs.randint(0, 2000, size=(1024, 1024)).astype(np.int32))
acc = {}
import torch
def wrap(obj, name, key):
    orig = getattr(obj, name); acc[key] = 0.0
    def timed(*a, **k):
        t0 = time.perf_counter(); r = orig(*a, **k)
        if key.startswith("_image") or key.startswith("som_device"): torch.cuda.synchronize()
        acc[key] += time.perf_counter() - t0; return r
    setattr(obj, name, timed)
wrap(pp, "_fov_device_half", "device half (caller's thread)")
wrap(pp, "_fov_table_half", "table half (finisher thread)")
wrap(pp, "_assemble_tables", "DataFrames + sample (finisher thread)")
wrap(flowsom, "fov_pixel_rows", "device rows (blur, filter, normalise, quantile, D2H)")
wrap(pp, "_read_segmentation", "read segmentation")
wrap(flowsom, "positive_quantile_f32", "positive_quantile_f32")
wrap(flowsom, "_image_to_device", "_image_to_device")
from ark_analysis_amd import som_device
wrap(som_device, "quantile_f32", "som_device.quantile_f32")
wrap(flowsom, "total_intensity_quantile_f32", "total_intensity_quantile_f32")
wrap(image_io, "read_channel", "read_channel (summed over threads)")
wrap(fov_tables, "write_dataframe", "write_dataframe (writer thread)")
wrap(image_io, "read_channels", "read_channels (all threads)")
wrap(pixel_cluster_utils, "calculate_channel_percentiles", "pass 1: channel percentiles")
wrap(pixel_cluster_utils, "calculate_pixel_intensity_percentile", "pass 2: pixel threshold")
W = fov_tables.TableWriter
for name in ("submit", "submit_call", "close"):
    wrap(W, name, "writer." + name)
pp.TableWriter = W
t0 = time.perf_counter()
pp.create_pixel_matrix(fovs, chans, root, tiff_dir, seg_dir)
total = time.perf_counter() - t0
print("total %.3f s for %d FOVs (%.1f ms/FOV, %.2f Mpx/s)" % (total, args.fovs, total / args.fovs * 1e3, args.fovs * 1.048576 / total))
print({k: round(v, 3) for k, v in acc.items()})
shutil.rmtree(root)
