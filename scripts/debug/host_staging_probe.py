"""What each host-side step of labelling one 1024^2 x 22 float64 FOV table costs (page-locked vs pageable)."""
import time
import numpy as np
import torch

n, c = 1 << 20, 22
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)


def t(fn, reps=4):
    out = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) * 1e3)
    return " ".join("%.1f" % v for v in out), r


print("pinned alloc [c,n] f64 (fresh each time, kept):", t(lambda: torch.empty((c, n), dtype=torch.float64, pin_memory=True))[0])
keep = []
def alloc_free():
    x = torch.empty((c, n), dtype=torch.float64, pin_memory=True); return None
print("pinned alloc + free (cache reuse):", t(alloc_free)[0])
cols = [np.random.rand(n) for _ in range(c)]
pin = torch.empty((c, n), dtype=torch.float64, pin_memory=True)
def gather():
    for j in range(c):
        pin[j].copy_(torch.from_numpy(cols[j]))
print("gather 22 columns -> pinned:", t(gather)[0])
pn = pin.numpy()
def gather_np():
    for j in range(c):
        np.copyto(pn[j], cols[j])
print("gather via numpy:", t(gather_np)[0])
d = torch.empty((c, n), dtype=torch.float64, device=dev)
print("H2D pinned block:", t(lambda: d.copy_(pin, non_blocking=True))[0])
def h2d_pageable():
    for j in range(c):
        d[j].copy_(torch.from_numpy(cols[j]), non_blocking=True)
print("H2D 22 pageable columns:", t(h2d_pageable)[0])
print("D2H .cpu() fresh pageable:", t(lambda: d.cpu())[0])
print("D2H into pinned:", t(lambda: pin.copy_(d, non_blocking=True))[0])
page = torch.empty((c, n), dtype=torch.float64)
print("D2H into reused pageable:", t(lambda: page.copy_(d))[0])
print("transpose [c,n]->[n,c] on device:", t(lambda: d.t().contiguous())[0])
