"""assign() throughput over a grid of (channels, nodes): looking for outliers."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from ark_analysis_amd import som_device as sd, synth
gpu = torch.device("cuda:0")
n = 1 << 20
print("rows/s in G (1 M fp32 rows); exact-path rows in parentheses")
for c in (8, 22, 40, 64, 100, 128):
    x = synth.make_fov_torch(n, c, seed=2, device=gpu)
    line = "C=%3d " % c
    for k in (64, 100, 144, 256, 400, 1024):
        w = x[torch.randperm(n, device=gpu)[:k]].double().contiguous()
        ws = sd.AssignWorkspace(n, c, k, gpu)
        labels = torch.empty(n, dtype=torch.int32, device=gpu)
        for _ in range(2): sd.assign(x, w, labels=labels, workspace=ws)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): sd.assign(x, w, labels=labels, workspace=ws)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        line += " K=%4d: %5.2f (%6d)" % (k, n / dt / 1e9, sd.last_exact_rows(ws))
    print(line)
