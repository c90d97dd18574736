"""Is the wave-private K8 kernel (config 2: 0.21 ms) held back by its occupancy (two waves per SIMD, LDS-limited by the
17.6 KB table per wave)?  Same rows, smaller tables (fewer clusters): more waves fit."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ark_analysis_amd import som_device
dev = torch.device("cuda")
n, c = 10 * 1024 * 1024, 22
x = torch.rand((n, c), device=dev, dtype=torch.float32)
for k in (100, 50, 25):
    labels = torch.randint(1, k + 1, (n,), device=dev, dtype=torch.int32)
    ts = []
    for rep in range(6):
        sums = torch.zeros((k, c), dtype=torch.float64, device=dev); counts = torch.zeros(k, dtype=torch.int64, device=dev)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        som_device.cluster_sums(x, labels, k, sums, counts)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("k = %3d: %.3f ms" % (k, sorted(ts)[1]))
