import sys, os
sys.path.insert(0, os.getcwd())
import torch
from ark_analysis_amd import som_device as sd
gpu = torch.device("cuda:0")
torch.manual_seed(0)
n, c, k = 40000, 22, 100
def ref(x, labels):
    s = torch.zeros((k, c), dtype=torch.float64, device=gpu); s.index_add_(0, (labels - 1).long(), x.double()); return s
x = torch.rand((n, c), device=gpu)
for name, labels in (("cyclic", (torch.arange(n, device=gpu) % k + 1).to(torch.int32)),
                     ("pairs equal", (torch.arange(n, device=gpu) // 2 % k + 1).to(torch.int32)),
                     ("random", torch.randint(1, k + 1, (n,), device=gpu, dtype=torch.int32))):
    s, cnt = sd.cluster_sums(x, labels, k)
    d = (s - ref(x, labels)).abs()
    print(name, "max abs diff %.3e" % d.max().item(), "bad entries", int((d > 1e-9).sum()), "total sum diff %.3e" % (s.sum() - x.double().sum()).item())
