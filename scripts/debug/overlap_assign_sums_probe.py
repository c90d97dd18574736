"""Does the mean-table pass (pxsom_cluster_sums, bound by LDS read-modify-write latency) hide under the next chunk's
BMU search (pxsom_assign, bound by MFMA + VALU issue) when the two run on different HIP streams?  Config 2 rows."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ark_analysis_amd import som_device, synth

dev = torch.device("cuda", 0)
n, c, k = 10 * 1024 * 1024, 22, 100
x = torch.cat([synth.make_fov_torch(1024 * 1024, c, 1000 + i, dev, torch.float32) for i in range(10)])
w = x[torch.randperm(n, device=dev)[:k]].double().contiguous()
labels = torch.empty(n, dtype=torch.int32, device=dev)
sums = torch.zeros((k, c), dtype=torch.float64, device=dev)
counts = torch.zeros(k, dtype=torch.int64, device=dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

def run(chunks, overlap):
    sums.zero_(); counts.zero_()
    bounds = [n * i // chunks for i in range(chunks + 1)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(chunks):
        a, b = bounds[i], bounds[i + 1]
        with torch.cuda.stream(sa):
            som_device.assign(x[a:b], w, labels=labels[a:b], workspace=ws[i])
            ev = torch.cuda.Event(); ev.record(sa)
        with torch.cuda.stream(sb if overlap else sa):
            if overlap:
                sb.wait_event(ev)
            som_device.cluster_sums(x[a:b], labels[a:b], k, sums, counts)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3

for chunks in (1, 2, 4, 5, 10, 20):
    ws = [som_device.AssignWorkspace((n + chunks - 1) // chunks + 1, c, k, dev) for _ in range(chunks)]
    for overlap in (False, True):
        run(chunks, overlap)
        ts = sorted(run(chunks, overlap) for _ in range(7))
        print("chunks %2d %s: %.3f ms (min %.3f)" % (chunks, "two streams" if overlap else "one stream ", ts[3], ts[0]))
    ref = counts.clone()
print("counts total", int(ref.sum()))
