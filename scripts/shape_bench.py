"""Assign + batch-train throughput of the non-headline BASELINE.json shapes (generic kernel paths)."""
import os, sys, json, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ark_analysis_amd import _capi, som_device, synth
from ark_analysis_amd.distributed import BatchSOMTrainer

dev = torch.device("cuda:0")
for name, n, c, xd, yd, dt in [("cfg4 cell SOM 1e6 x 100, K=100", 1_000_000, 100, 10, 10, torch.float32),
                               ("cfg5 shape 2048^2 x 40 fp32, K=400", 4 * 1024 * 1024, 40, 20, 20, torch.float32),
                               ("cfg5 2048^2 x 40 fp16, K=400", 4 * 1024 * 1024, 40, 20, 20, torch.float16),
                               ("cfg1 shape 512^2 x 8, K=100", 512 * 512, 8, 10, 10, torch.float32)]:
    k = xd * yd
    x = synth.make_fov_torch(n, c, seed=3, device=dev).to(dt)
    esz = x.element_size()
    w = x[torch.randperm(n, device=dev)[:k]].double().contiguous()
    xt = x[::10].contiguous()
    BatchSOMTrainer(xd, yd, c, dev, batch_steps=16).train(xt, w, 1)
    trainer = BatchSOMTrainer(xd, yd, c, dev, batch_steps=64)
    w_t = w.clone()
    trainer.train(xt, w_t, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        trainer.train(xt, w_t, 1)
    torch.cuda.synchronize()
    train_ms = (time.perf_counter() - t0) / 3 * 1e3
    ws = som_device.AssignWorkspace(n, c, k, dev)
    labels = torch.empty(n, dtype=torch.int32, device=dev)
    for _ in range(2):
        som_device.assign(x, w, labels=labels, workspace=ws)
    torch.cuda.synchronize()
    t = _capi.KernelTimer(min_rows=n)
    t0 = time.perf_counter()
    with t:
        for _ in range(5):
            som_device.assign(x, w, labels=labels, workspace=ws)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 5
        ms, cnt = t.collect()
    print(json.dumps({"shape": name, "train_pass_ms_64_steps_10pct": round(train_ms, 3), "filter_ms": round(ms / cnt, 3), "assign_wall_ms": round(wall * 1e3, 3),
                      "Mpx_per_s": round(n / wall / 1e6, 1), "GBps_algorithmic": round((c * esz + 4) * n / (ms / cnt) / 1e6, 1),
                      "exact_rows": som_device.last_exact_rows(ws)}))
