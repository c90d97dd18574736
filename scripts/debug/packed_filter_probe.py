"""Runs the packed-K filter (config 5 shape: 2048^2 x 40 binary16 rows, 20 x 20 nodes) a few times: target of rocprofv3 counter passes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ark_analysis_amd import som_device, synth

dev = torch.device("cuda:0")
n, c, k = 4 * 1024 * 1024, 40, 400
x = synth.make_fov_torch(n, c, seed=3, device=dev).to(torch.float16)
w = x[torch.randperm(n, device=dev)[:k]].double().contiguous()
ws = som_device.AssignWorkspace(n, c, k, dev)
labels = torch.empty(n, dtype=torch.int32, device=dev)
for _ in range(6):
    som_device.assign(x, w, labels=labels, workspace=ws)
torch.cuda.synchronize()
