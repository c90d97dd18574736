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
from ark_analysis_amd.distributed import BatchSOMTrainer
BatchSOMTrainer(10, 10, C, dev, batch_steps=64).train(x[::10].contiguous(), w, 1)     # a trained codebook, as in bench.py
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
ws2 = sd.AssignSumsWorkspace(n, C, K, dev)
def one():
    sums.zero_(); counts.zero_()
    sd.assign_sums(x, w, labels=labels, sums=sums, counts=counts, workspace=ws2)
print("assign + cluster_sums: %.3f ms" % timeit(two))
print("accumulating filter  : %.3f ms" % timeit(one))
two(); s2 = sums.clone(); c2 = counts.clone(); l2 = labels.clone()
one()
print("labels equal", bool(torch.equal(l2, labels)), "sums max rel diff", float(((s2 - sums).abs() / s2.abs().clamp(min=1e-300)).max()),
      "counts equal", bool(torch.equal(c2, counts)))
