"""Labels + mean table in one pass (pxsom_assign_means) when neighbouring rows share their label -- what real images do and the
synthetic FOVs hide: the 16 rows of one LDS atomic instruction then hit the same table words.  Same rows, three orders:
as generated, sorted by label (every 64-row group one label), and in runs of R equal labels (image-like)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ark_analysis_amd import synth, som_device as sd

dev = torch.device("cuda:0")
n, c, k = 10 << 20, 22, 100
x = synth.make_fov_torch(n, c, seed=7, device=dev)
if len(sys.argv) > 1 and sys.argv[1] == "f64":     # what the drop-in classes hold (the two-tile kernel, pxsom_assign_onepass.h)
    n = 4 << 20
    x = x[:n].double().contiguous()
w = x[torch.randperm(n, device=dev)[:k]].double().contiguous()
for _ in range(3):   # a few Lloyd steps: a codebook like a trained one
    lab, _ = sd.assign(x, w)
    s, cnt = sd.cluster_sums(x, lab, k)
    w = (s / cnt.clamp(min=1).double().unsqueeze(1)).contiguous()
lab, _ = sd.assign(x, w)
order = torch.argsort(lab.long(), stable=True)


def runs(r):
    """rows in runs of r equal labels: the sorted order cut into pieces of r rows, the pieces shuffled"""
    pieces = n // r
    perm = torch.randperm(pieces, device=dev)
    idx = (perm.unsqueeze(1) * r + torch.arange(r, device=dev).unsqueeze(0)).reshape(-1)
    return order[idx]


def timed(xx, fn_name):
    m = xx.shape[0]                  # (runs of a length that does not divide n leave a few rows out)
    ws = sd.AssignSumsWorkspace(m, c, k, dev)
    labels = torch.empty(m, dtype=torch.int32, device=dev)
    sums = torch.empty((k, c), dtype=torch.float64, device=dev); counts = torch.empty(k, dtype=torch.int64, device=dev); means = torch.empty_like(sums)
    if fn_name == "cluster_sums":
        sd.assign(xx, w, labels=labels, workspace=ws)
    fn = {"assign_means": lambda: sd.assign_means(xx, w, labels, sums, counts, means, ws),
          "assign": lambda: sd.assign(xx, w, labels=labels, workspace=ws),
          "cluster_sums": lambda: sd.cluster_sums(xx, labels, k, sums=sums, counts=counts)}[fn_name]
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 10 * 1e3, means.clone()


base_ms, base_means = timed(x, "assign_means")
print("rows as generated:            labels + mean table %.3f ms, labels only %.3f ms, sums kernel alone %.3f ms" % (base_ms, timed(x, "assign")[0], timed(x, "cluster_sums")[0]))
for name, idx in (("sorted by label", order), ("runs of 256 equal labels", runs(256)), ("runs of 64", runs(64)), ("runs of 16", runs(16)), ("runs of 8", runs(8)), ("runs of 6", runs(6)), ("runs of 5", runs(5)), ("runs of 4", runs(4)), ("runs of 2", runs(2))):
    xs = x[idx].contiguous()
    ms, means = timed(xs, "assign_means")
    ok = torch.allclose(means, base_means, rtol=1e-9, atol=0) if xs.shape[0] == n else "n/a (a few rows left out)"
    print("%-28s  labels + mean table %.3f ms, labels only %.3f ms, sums kernel alone %.3f ms  (means equal to the unsorted run's within 1e-9: %s)"
          % (name + ":", ms, timed(xs, "assign")[0], timed(xs, "cluster_sums")[0], ok))
    del xs
