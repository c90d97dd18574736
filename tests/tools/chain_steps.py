import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from ark_analysis_amd import som_device as sd, synth
from ark_analysis_amd.distributed import batch_schedule, BatchSOMTrainer
from ark_analysis_amd.flowsom import default_radius_range
from tests import oracle_binding as ob
gpu = torch.device("cuda:0")
xdim, ydim, c, dtype, M = 10, 10, 16, np.float64, 4
k, n = 100, 9000
x = synth.make_fov_numpy(n, c, seed=41, dtype=dtype)
rs = np.random.RandomState(8)
w = np.ascontiguousarray(x[rs.choice(n, k, replace=False)].astype(np.float64))
xd = torch.from_numpy(x).to(gpu)
rr = default_radius_range(xdim, ydim)
want = ob.som_batch(x.astype(np.float64), w, xdim, ydim, 1, (0.05, 0.01), rr, M)

def manual(sync, ws_n, fresh_labels):
    wd = torch.from_numpy(w.copy()).to(gpu)
    stats = torch.empty(k * (c + 1), dtype=torch.float64, device=gpu)
    ws = sd.AssignWorkspace(ws_n, c, k, gpu)
    lab = torch.empty(n, dtype=torch.int32, device=gpu)
    for g in range(M):
        xv = xd[g::M]
        labels = torch.empty(xv.shape[0], dtype=torch.int32, device=gpu) if fresh_labels else lab
        sd.batch_accumulate(xv, wd, labels, stats, ws)
        if sync: torch.cuda.synchronize()
        thr, alpha = batch_schedule(g, M, (0.05, 0.01), rr)
        sd.batch_update(wd, xdim, ydim, stats[:k * c].view(k, c), stats[k * c:], thr, alpha)
        if sync: torch.cuda.synchronize()
    return np.abs(wd.cpu().numpy() / want - 1).max()

for sync in (True, False):
    for ws_n in (9000, 2250):
        for fresh in (True, False):
            print("manual sync", sync, "ws_n", ws_n, "fresh labels", fresh, ["%.2e" % manual(sync, ws_n, fresh) for _ in range(3)], flush=True)
tr = BatchSOMTrainer(xdim, ydim, c, gpu, batch_steps=M)
for rep in range(3):
    wb = torch.from_numpy(w.copy()).to(gpu)
    tr.train(xd, wb, 1)
    print("trainer", "%.2e" % np.abs(wb.cpu().numpy() / want - 1).max())
