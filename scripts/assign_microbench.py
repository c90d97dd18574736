"""Times pxsom_assign's filter kernel alone on the BASELINE config-2 matrix (HIP events)."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ark_analysis_amd import _capi, som_device, synth

def main():
    dev = torch.device("cuda:0")
    F, P, C, K = int(os.environ.get("FOVS", 10)), 1024 * 1024, 22, 100
    n = F * P
    x = torch.empty((n, C), dtype=torch.float32, device=dev)
    for f in range(F):
        x[f * P:(f + 1) * P] = synth.make_fov_torch(P, C, seed=1000 + f, device=dev)
    w = x[torch.randperm(n, device=dev)[:K]].double().contiguous()
    if os.environ.get("TRAINED", "0") == "1":
        from ark_analysis_amd.distributed import BatchSOMTrainer
        BatchSOMTrainer(10, 10, C, dev, batch_steps=64).train(x[::10].contiguous(), w, 1)
    ws = som_device.AssignWorkspace(n, C, K, dev)
    labels = torch.empty(n, dtype=torch.int32, device=dev)
    for _ in range(3):
        som_device.assign(x, w, labels=labels, workspace=ws)
    torch.cuda.synchronize()
    t = _capi.KernelTimer(min_rows=n)
    with t:
        for _ in range(10):
            som_device.assign(x, w, labels=labels, workspace=ws)
        ms, cnt = t.collect()
    avg = ms / cnt
    print(json.dumps({"mode": os.environ.get("PXSOM_FILTER_MODE", "0"), "filter_ms": round(avg, 4),
                      "GBps": round(92 * n / avg / 1e6, 1), "exact_rows": som_device.last_exact_rows(ws), "trained": os.environ.get("TRAINED", "0")}))

main()
