"""VERDICT lever "contiguous mini-batches": what a training step costs on the strided rows i = t (mod 64) of the
training subset against the same rows stored contiguously (config 2: the one-launch step kernel; config 5 at full
size: four launches per step)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ark_analysis_amd import som_device
dev = torch.device("cuda")

def run(name, n_train, c, xd, yd, dt, reps=30):
    x = torch.rand((n_train, c), device=dev, dtype=torch.float32).to(dt)
    w0 = x[torch.randperm(n_train, device=dev)[: xd * yd]].double().contiguous()
    block = x[0::64].contiguous()                       # mini-batch 0, stored contiguously
    out = {}
    for label, mat, m in (("strided", x, 64), ("contiguous", block, 1)):
        st = som_device.BatchTrainState(mat.shape[0], c, xd, yd, m, dev)
        st.wbuf[0].copy_(w0)
        ts = []
        for rep in range(reps):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            som_device.batch_train_steps(mat, st, 0, 1, 64, (0.05, 0.01), (6.0, 0.0))   # step 0 of 64
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e6)
        out[label] = sorted(ts)[len(ts) // 4]
    print("%s: one step on %d rows: strided %.1f us, contiguous %.1f us (host-timed, launch included)" % (
        name, block.shape[0], out["strided"], out["contiguous"]))

run("config 2 (10x10, 22 ch fp32)", 1 << 20, 22, 10, 10, torch.float32)
run("config 3 (10x10, 22 ch fp32, 25 FOVs)", 2621440, 22, 10, 10, torch.float32)
run("config 5, 4 FOVs (20x20, 40 ch fp16)", 1677722, 40, 20, 20, torch.float16)
run("config 5, 62 FOVs", 26004685, 40, 20, 20, torch.float16)
