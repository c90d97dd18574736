"""Exact-online training speed per shape (us / step)."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from ark_analysis_amd import som_device as sd, synth
from ark_analysis_amd.flowsom import default_radius_range
gpu = torch.device("cuda:0")
for (c, xd, yd) in [(22, 10, 10), (40, 10, 10), (100, 10, 10), (60, 10, 10), (100, 8, 8), (40, 20, 20), (100, 12, 12)]:
    n = 200_000
    x = synth.make_fov_torch(n, c, seed=1, device=gpu)
    k = xd * yd
    w = x[torch.randperm(n, device=gpu)[:k]].double().contiguous()
    order = torch.randint(0, n, (n,), device=gpu, dtype=torch.int64)
    rr = default_radius_range(xd, yd)
    sd.train_online(x, w.clone(), xd, yd, 1, (0.05, 0.01), rr, order)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sd.train_online(x, w.clone(), xd, yd, 1, (0.05, 0.01), rr, order)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("C=%3d %2dx%2d: %.3f us/step" % (c, xd, yd, dt / n * 1e6))
