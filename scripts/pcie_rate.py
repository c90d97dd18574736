"""PCIe-inclusive rate of the host-buffer boundary (flowsom.map_data_to_nodes): host f64 table in,
labels + distances out.  Reported in DESIGN.md; never the bench `value`."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ark_analysis_amd import flowsom, synth
import torch

x = synth.make_fov_numpy(1 << 20, 22, seed=1, dtype=np.float64)
w = x[np.random.RandomState(0).choice(len(x), 100, replace=False)].copy()
flowsom.map_data_to_nodes(w, x[:1000])
torch.cuda.synchronize()
best = 1e9
for _ in range(3):
    t = time.perf_counter()
    labels, dists = flowsom.map_data_to_nodes(w, x)
    best = min(best, time.perf_counter() - t)
x32 = x.astype(np.float32)
b32 = 1e9
for _ in range(3):
    t = time.perf_counter()
    labels, dists = flowsom.map_data_to_nodes(w, x32)
    b32 = min(b32, time.perf_counter() - t)
print(json.dumps({"rows": len(x), "f64_host_to_host_ms": round(best * 1e3, 2), "Mpx_per_s_f64": round(len(x) / best / 1e6, 1),
                  "f32_host_to_host_ms": round(b32 * 1e3, 2), "Mpx_per_s_f32": round(len(x) / b32 / 1e6, 1)}))
