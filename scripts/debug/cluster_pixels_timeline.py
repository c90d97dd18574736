"""Where cluster_pixels' wall time goes: waits on the readers, labelling on the caller's thread, waits on the
writers (queue full), final drain.  Builds N synthetic 1024^2 x 22 tables like scripts/pipeline_bench.py."""
import argparse, os, shutil, sys, tempfile, time
import numpy as np, pandas as pd, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ark_analysis_amd import fov_tables, synth, arrow_assign
from ark_analysis_amd.phenotyping import pixel_som_clustering as psc

ap = argparse.ArgumentParser(); ap.add_argument("--fovs", type=int, default=12); args = ap.parse_args()
root = tempfile.mkdtemp(prefix="pxsom_tl_")
chans = ["chan%d" % i for i in range(22)]; fovs = ["fov%d" % i for i in range(args.fovs)]
n = 1 << 20
os.mkdir(os.path.join(root, "pixel_mat_data")); os.mkdir(os.path.join(root, "pixel_mat_subsetted"))
for i, fov in enumerate(fovs):
    x = synth.make_fov_numpy(n, 22, seed=1000 + i, dtype=np.float64)
    df = pd.DataFrame(x, columns=chans); df["fov"] = fov
    df["row_index"] = np.repeat(np.arange(1024), 1024); df["column_index"] = np.tile(np.arange(1024), 1024); df["label"] = 0
    fov_tables.write_dataframe(df, os.path.join(root, "pixel_mat_data", fov + ".feather"))
    fov_tables.write_dataframe(df.iloc[::10], os.path.join(root, "pixel_mat_subsetted", fov + ".feather"))
fov_tables.write_dataframe(pd.DataFrame(np.ones((1, 22)), columns=chans), os.path.join(root, "post_rowsum_chan_norm.feather"))
som = psc.train_pixel_som(fovs, chans, root)
acc = {"label": 0.0, "submit": 0.0, "close": 0.0, "stage": 0.0, "read": 0.0, "write": 0.0}
orig_label = psc._label_table
per_fov = []
def timed_label(*a, **k):
    t0 = time.perf_counter(); r = orig_label(*a, **k); dt = time.perf_counter() - t0; acc["label"] += dt
    per_fov.append(round(dt * 1e3, 1)); return r
psc._label_table = timed_label
W = psc.TableWriter
orig_submit, orig_close = W.submit, W.close
def submit(self, *a, **k):
    t0 = time.perf_counter(); orig_submit(self, *a, **k); acc["submit"] += time.perf_counter() - t0
def close(self):
    t0 = time.perf_counter(); orig_close(self); acc["close"] += time.perf_counter() - t0
W.submit, W.close = submit, close
orig_write = fov_tables.write_dataframe
def write(*a, **k):
    t0 = time.perf_counter(); orig_write(*a, **k); acc["write"] += time.perf_counter() - t0
fov_tables.write_dataframe = write
orig_load = fov_tables.FovTableDir.load_arrow
def load(self, fov):
    t0 = time.perf_counter(); r = orig_load(self, fov); acc["read"] += time.perf_counter() - t0; return r
fov_tables.FovTableDir.load_arrow = load
def wrap(obj, name, key):
    orig = getattr(obj, name)
    acc[key] = 0.0
    def timed(*a, **k):
        t0 = time.perf_counter(); r = orig(*a, **k); acc[key] += time.perf_counter() - t0; return r
    setattr(obj, name, timed)
wrap(fov_tables.FovTableDir, "commit", "commit")
wrap(psc.pixel_cluster_utils, "find_fovs_missing_col", "missing_col")
wrap(psc, "_check_columns_against", "check_columns")
orig_iter = fov_tables.TablePrefetcher.__iter__
def timed_iter(self):
    it = orig_iter(self)
    while True:
        t0 = time.perf_counter()
        try:
            item = next(it)
        except StopIteration:
            return
        acc["wait_reader"] = acc.get("wait_reader", 0.0) + time.perf_counter() - t0
        yield item
fov_tables.TablePrefetcher.__iter__ = timed_iter
t0 = time.perf_counter()
psc.cluster_pixels(fovs, root, som)
total = time.perf_counter() - t0
print("total %.3f s for %d FOVs (%.1f ms/FOV); caller thread: label %.3f, submit waits %.3f, final drain %.3f; "
      "summed over threads: read %.3f, write %.3f" % (total, args.fovs, total / args.fovs * 1e3, acc["label"],
      acc["submit"], acc["close"], acc["read"], acc["write"]))
print({k: round(v, 3) for k, v in acc.items()})
print('label ms per FOV:', per_fov)
fov_tables.wait_for_cleanup()
shutil.rmtree(root)
