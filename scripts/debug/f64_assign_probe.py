"""Labels (pxsom_assign) and labels + mean table (pxsom_assign_means) on binary64 rows -- what the drop-in classes hold -- against
binary32 rows of the same values: ms per call on 4 x 1024^2 x 22."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ark_analysis_amd import synth, som_device as sd

dev = torch.device("cuda:0")
n, c, k = 4 << 20, 22, 100
x32 = synth.make_fov_torch(n, c, seed=7, device=dev)
w = x32[torch.randperm(n, device=dev)[:k]].double().contiguous()
for _ in range(3):   # a few Lloyd steps: a codebook like a trained one
    lab, _ = sd.assign(x32, w)
    s, cnt = sd.cluster_sums(x32, lab, k)
    w = (s / cnt.clamp(min=1).double().unsqueeze(1)).contiguous()
for x in (x32, x32.double().contiguous()):
    ws = sd.AssignSumsWorkspace(n, c, k, dev)
    labels = torch.empty(n, dtype=torch.int32, device=dev)
    sums = torch.empty((k, c), dtype=torch.float64, device=dev); counts = torch.empty(k, dtype=torch.int64, device=dev); means = torch.empty_like(sums)
    for name, fn in (("assign", lambda: sd.assign(x, w, labels=labels, workspace=ws)), ("assign_means", lambda: sd.assign_means(x, w, labels, sums, counts, means, ws))):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): fn()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        print("%s %-12s %.3f ms  (%.2f TB/s of rows)" % (str(x.dtype).split(".")[1], name, ms, n * c * x.element_size() / ms / 1e9))
