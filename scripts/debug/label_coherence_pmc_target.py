"""Target of the LDS-counter passes of scripts/jobs/r5_coherence_pmc.sh: the one-pass kernel on rows sorted by label (argv[1] = sorted) or
as generated, six launches."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ark_analysis_amd import synth, som_device as sd

dev = torch.device("cuda:0")
n, c, k = 10 << 20, 22, 100
x = synth.make_fov_torch(n, c, seed=7, device=dev)
w = x[torch.randperm(n, device=dev)[:k]].double().contiguous()
lab, _ = sd.assign(x, w)
if len(sys.argv) > 1 and sys.argv[1] == "sorted":
    x = x[torch.argsort(lab.long(), stable=True)].contiguous()
ws = sd.AssignSumsWorkspace(n, c, k, dev)
labels = torch.empty(n, dtype=torch.int32, device=dev)
sums = torch.empty((k, c), dtype=torch.float64, device=dev); counts = torch.empty(k, dtype=torch.int64, device=dev); means = torch.empty_like(sums)
for _ in range(6):
    sd.assign_means(x, w, labels, sums, counts, means, ws)
torch.cuda.synchronize()
