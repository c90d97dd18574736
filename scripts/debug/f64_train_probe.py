"""Training pass of the pixel SOM on binary64 rows (what the drop-in classes hold: feather tables are float64): ms per pass on
1 048 576 x 22, default schedule.  PXSOM_STEP_TPW forces the tiles per wave of every fused step (1 / 2 / 4)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ark_analysis_amd import synth
from ark_analysis_amd.distributed import BatchSOMTrainer

dev = torch.device("cuda:0")
for dt in (torch.float64, torch.float32):
    x = synth.make_fov_torch(1 << 20, 22, seed=7, device=dev).to(dt).contiguous()
    w0 = x[torch.randperm(x.shape[0], device=dev)[:100]].double().contiguous()
    tr = BatchSOMTrainer(10, 10, 22, dev)
    for _ in range(3):
        tr.train(x, w0.clone(), num_passes=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        tr.train(x, w0.clone(), num_passes=1)
    torch.cuda.synchronize()
    print("%s rows, PXSOM_STEP_TPW=%s: %.3f ms per pass" % (str(dt).split(".")[1], os.environ.get("PXSOM_STEP_TPW", "auto"), (time.perf_counter() - t0) / 20 * 1e3))
