"""Is the one-pass labels + tables kernel waiting for HBM?  Its time per row on matrices that stay in the L2s / the MALL against
config 2's 923 MB (same codebook, same kernel; HIP events around back-to-back calls)."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ark_analysis_amd import som_device, synth
from ark_analysis_amd.distributed import BatchSOMTrainer
dev = torch.device("cuda:0")
C, K, P = 22, 100, 1024 * 1024
x = torch.cat([synth.make_fov_torch(P, C, seed=1000 + f, device=dev) for f in range(10)])
w = x[torch.randperm(x.shape[0], device=dev)[:K]].double().contiguous()
BatchSOMTrainer(10, 10, C, dev).train(x[::10].contiguous(), w, 1)
for n in (65536, 262144, 1 << 20, 4 << 20, 10 << 20):
    xs = x[:n]
    wsum = som_device.AssignSumsWorkspace(n, C, K, dev)
    ws = som_device.AssignWorkspace(n, C, K, dev)
    labels = torch.empty(n, dtype=torch.int32, device=dev)
    sums = torch.zeros((K, C), dtype=torch.float64, device=dev); counts = torch.zeros(K, dtype=torch.int64, device=dev)
    out = {}
    for name, fn in (("labels+tables", lambda: som_device.assign_sums(xs, w, labels=labels, sums=sums, counts=counts, workspace=wsum)),
                     ("labels only", lambda: som_device.assign(xs, w, labels=labels, workspace=ws))):
        for _ in range(3): fn()
        reps = max(5, min(200, (40 << 20) // n))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        out[name] = {"ms_per_call": round(ms, 4), "ns_per_row_chip": round(ms * 1e6 / n, 4), "Gpx_s": round(n / ms / 1e6, 2)}
    print(json.dumps({"rows": n, "MB": round(n * C * 4 / 1e6, 1), **out}))
