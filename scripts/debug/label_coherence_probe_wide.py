"""The two-pass mean table of the wide shapes (config 5: binary16 rows x 40 channels, 400 clusters -- the atomic sums kernel with one
128 KB table per CU; config 4: binary32 x 100 columns, 100 clusters) on rows whose neighbours share their label."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ark_analysis_amd import synth, som_device as sd

dev = torch.device("cuda:0")
for n, c, k, dt in ((16 << 20, 40, 400, torch.float16), (2 << 20, 100, 100, torch.float32)):
    x = synth.make_fov_torch(n, c, seed=7, device=dev).to(dt)
    w = x[torch.randperm(n, device=dev)[:k]].double().contiguous()
    lab, _ = sd.assign(x, w)
    order = torch.argsort(lab.long(), stable=True)

    def runs(r):
        pieces = n // r
        perm = torch.randperm(pieces, device=dev)
        return order[(perm.unsqueeze(1) * r + torch.arange(r, device=dev).unsqueeze(0)).reshape(-1)]

    def timed(xx, ll):
        sums = torch.empty((k, c), dtype=torch.float64, device=dev); counts = torch.empty(k, dtype=torch.int64, device=dev)
        for _ in range(2): sd.cluster_sums(xx, ll, k, sums=sums, counts=counts)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): sd.cluster_sums(xx, ll, k, sums=sums, counts=counts)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 5 * 1e3

    print("%d x %d %s rows, %d clusters: sums kernel on rows as generated %.3f ms" % (n, c, str(dt).split(".")[1], k, timed(x, lab)))
    for name, idx in (("sorted by label", order), ("runs of 64", runs(64)), ("runs of 16", runs(16)), ("runs of 4", runs(4))):
        xs, ls = x[idx].contiguous(), lab[idx].contiguous()
        print("   %-18s %.3f ms" % (name + ":", timed(xs, ls)))
        del xs, ls
    del x, lab, order
