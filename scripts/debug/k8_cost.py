import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from ark_analysis_amd import som_device as sd, synth, _capi
gpu = torch.device("cuda:0")
P, C, K = 1024 * 1024, 22, 100
x = torch.empty((10 * P, C), dtype=torch.float32, device=gpu)
for f in range(10):
    x[f * P:(f + 1) * P] = synth.make_fov_torch(P, C, seed=1000 + f, device=gpu)
n = x.shape[0]
w = x[torch.randperm(n, device=gpu)[:K]].double().contiguous()
labels = torch.empty(n, dtype=torch.int32, device=gpu)
stats = torch.empty(K * (C + 1), dtype=torch.float64, device=gpu)
ws = sd.AssignWorkspace(n, C, K, gpu)
def t(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
a = t(lambda: sd.assign(x, w, labels=labels, workspace=ws))
s1, c1 = sd.cluster_sums(x, labels, K)
b = t(lambda: sd.cluster_sums(x, labels, K))
c = t(lambda: sd.batch_accumulate(x, w, labels, stats, ws))
timer = _capi.KernelTimer(min_rows=n)
with timer:
    for _ in range(5): sd.batch_accumulate(x, w, labels, stats, ws)
    torch.cuda.synchronize()
    ms, nl = timer.collect()
print("assign %.3f ms, cluster_sums %.3f ms, fused assign+sums %.3f ms (filter kernel %.3f ms)" % (a, b, c, ms / max(nl, 1)))
got = stats.cpu().numpy()
print("fused vs separate: counts equal", np.array_equal(got[K * C:], c1.cpu().numpy().astype(np.float64)),
      "sums maxrel %.2e" % np.abs(got[:K * C].reshape(K, C) / s1.cpu().numpy() - 1).max())
