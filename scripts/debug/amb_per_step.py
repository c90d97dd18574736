import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from ark_analysis_amd import som_device as sd, synth
from ark_analysis_amd.distributed import BatchSOMTrainer
gpu = torch.device("cuda:0")
P, C, K = 1024 * 1024, 22, 100
x = torch.cat([synth.make_fov_torch(P, C, seed=1000 + f, device=gpu) for f in range(10)])[::10].contiguous()
g = torch.Generator(device="cpu"); g.manual_seed(42)
w = x[torch.randperm(x.shape[0], generator=g)[:K].to(gpu)].double().contiguous()
tr = BatchSOMTrainer(10, 10, C, gpu, batch_steps=64)
counts = []
for step in range(64):
    view = x[step::64]
    ws = sd.AssignWorkspace(view.shape[0], C, K, gpu)
    sd.assign(view, w, workspace=ws)
    counts.append(sd.last_exact_rows(ws))
    tr.step(x, w, step, 64, chained=step > 0)
print("listed rows per step (of %d):" % view.shape[0], counts)
