"""How long does the accumulating filter (labels + per-cluster sums in one pass) take over all rows of config 2,
against assign + cluster_sums as two passes?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ark_analysis_amd import som_device as sd, synth
dev = torch.device("cuda:0")
F, P, C, K = 10, 1024 * 1024, 22, 100
x = torch.cat([synth.make_fov_torch(P, C, seed=1000 + f, device=dev) for f in range(F)])
w = x[torch.randperm(F * P, device=dev)[:K]].double().contiguous()
n = x.shape[0]
labels = torch.empty(n, dtype=torch.int32, device=dev)
ws = sd.AssignWorkspace(n, C, K, dev)
stats = torch.zeros(K * (C + 1), dtype=torch.float64, device=dev)
sums = torch.zeros((K, C), dtype=torch.float64, device=dev)
counts = torch.zeros(K, dtype=torch.int64, device=dev)
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
def two():
    sd.assign(x, w, labels=labels, workspace=ws)
    sums.zero_(); counts.zero_()
    sd.cluster_sums(x, labels, K, sums=sums, counts=counts)
def one():
    sd.batch_accumulate(x, w, labels, stats, ws)
print("assign + cluster_sums: %.3f ms" % timeit(two))
print("accumulating filter  : %.3f ms" % timeit(one))
two(); s2 = sums.clone(); c2 = counts.clone(); l2 = labels.clone()
one()
print("labels equal", bool(torch.equal(l2, labels)), "sums equal", bool(torch.equal(s2.reshape(-1), stats[:K*C])),
      "counts equal", bool(torch.equal(c2.double(), stats[K*C:])))
