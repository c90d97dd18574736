"""Training pass of a cell SOM on tables of realistic size (cells x 100 pixel-cluster-count features, 10 x 10 nodes, default
schedule): ms per pass.  PXSOM_STEP_WIDE=0 switches the one-launch steps for wide codebooks off (launch-per-phase route)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ark_analysis_amd.distributed import BatchSOMTrainer

dev = torch.device("cuda:0")
for n in (20_000, 50_000, 200_000):
    g = torch.Generator(device=dev); g.manual_seed(5)
    x = torch.poisson(torch.full((n, 100), 3.0, device=dev), generator=g)
    x.div_(torch.empty((n, 1), device=dev).uniform_(50.0, 500.0, generator=g))
    x = (x / torch.quantile(x[: min(n, 1 << 20)], 0.999, dim=0).clamp(min=1e-9)).contiguous()
    w0 = x[torch.randperm(n, device=dev)[:100]].double().contiguous()
    tr = BatchSOMTrainer(10, 10, 100, dev)
    for _ in range(3):
        tr.train(x, w0.clone(), num_passes=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        tr.train(x, w0.clone(), num_passes=1)
    torch.cuda.synchronize()
    print("cells %7d  wide=%s  %.3f ms per pass (%d steps)" % (n, os.environ.get("PXSOM_STEP_WIDE", "1"), (time.perf_counter() - t0) / reps * 1e3,
                                                             tr.schedule.steps))
