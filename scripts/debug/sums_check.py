"""cluster_sums across shapes: equality with an fp64 index_add_ reference (fp32 rows sum exactly) + timing."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from ark_analysis_amd import som_device as sd
gpu = torch.device("cuda:0")
torch.manual_seed(0)

def ref(x, labels, k):
    ok = (labels >= 1) & (labels <= k)
    s = torch.zeros((k, x.shape[1]), dtype=torch.float64, device=gpu)
    s.index_add_(0, (labels[ok] - 1).long(), x[ok].double())
    cnt = torch.bincount((labels[ok] - 1).long(), minlength=k)
    return s, cnt

def t(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3

bad = 0
cases = [(4_194_304, 22, 100, torch.float32, "uniform"), (4_194_304, 22, 100, torch.float32, "runs"),
         (4_194_304, 22, 100, torch.float32, "skew"), (1_000_003, 22, 100, torch.float32, "uniform"),
         (4097, 22, 100, torch.float32, "uniform"), (65_536, 22, 100, torch.float32, "uniform"),
         (2_000_000, 40, 400, torch.float32, "uniform"), (2_000_000, 16, 100, torch.float32, "uniform"),
         (2_000_000, 21, 100, torch.float32, "uniform"), (2_000_000, 33, 64, torch.float32, "uniform"),
         (2_000_000, 64, 100, torch.float32, "uniform"), (2_000_000, 8, 100, torch.float32, "uniform"),
         (2_000_000, 3, 10, torch.float32, "uniform"), (2_000_000, 1, 5, torch.float32, "uniform"),
         (1_000_000, 22, 100, torch.float64, "uniform"), (1_000_000, 22, 100, torch.float16, "uniform"),
         (500_000, 22, 8, torch.float32, "uniform"), (500_000, 22, 1, torch.float32, "uniform"),
         (300_000, 70, 100, torch.float32, "uniform")]
for (n, c, k, dt, mode) in cases:
    x = (torch.rand((n, c), device=gpu) * 3).to(dt)
    if mode == "uniform":
        labels = torch.randint(0, k + 2, (n,), device=gpu, dtype=torch.int32)   # 0 and k+1 are skipped
    elif mode == "runs":
        labels = (torch.arange(n, device=gpu) // 3 % k + 1).to(torch.int32)      # equal neighbours: clashes
    else:
        labels = torch.where(torch.rand(n, device=gpu) < 0.7, 1, torch.randint(1, k + 1, (n,), device=gpu)).to(torch.int32)
    s, cnt = sd.cluster_sums(x, labels, k)
    rs, rc = ref(x, labels, k)
    okc = torch.equal(cnt.long(), rc.long())
    if dt == torch.float64:
        oks = torch.allclose(s, rs, rtol=1e-12, atol=1e-9)
    else:
        oks = torch.equal(s, rs) or torch.allclose(s, rs, rtol=1e-13, atol=0)
    ms = t(lambda: sd.cluster_sums(x, labels, k))
    gb = n * (c * x.element_size() + 4) / 1e9
    print("n=%8d c=%2d k=%3d %-8s %-7s counts %s sums %s  %.3f ms  %.0f GB/s" % (n, c, k, str(dt)[6:], mode, okc, oks, ms, gb / ms * 1e3))
    bad += (not okc) + (not oks)
# strided view and a row range not starting at 0
x = torch.rand((1_000_000, 30), device=gpu)
labels = torch.randint(1, 101, (1_000_000,), device=gpu, dtype=torch.int32)
v = x[:, 4:26]
s, cnt = sd.cluster_sums(v, labels, 100)
rs, rc = ref(v, labels, 100)
print("strided view:", torch.equal(cnt.long(), rc.long()), torch.allclose(s, rs, rtol=1e-13, atol=0))
bad += not (torch.equal(cnt.long(), rc.long()) and torch.allclose(s, rs, rtol=1e-13, atol=0))
print("FAILURES", bad)
