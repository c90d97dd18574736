import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from ark_analysis_amd import som_device as sd, synth
from ark_analysis_amd.distributed import BatchSOMTrainer, HipKernels
from ark_analysis_amd.flowsom import default_radius_range
from tests import oracle_binding as ob
gpu = torch.device("cuda:0")
for (xdim, ydim, c, dtype) in [(10, 10, 22, np.float32), (10, 10, 16, np.float64), (10, 10, 16, np.float32), (7, 9, 12, np.float32)]:
    k, n = xdim * ydim, 9000
    x = synth.make_fov_numpy(n, c, seed=41, dtype=dtype)
    rs = np.random.RandomState(8)
    w = np.ascontiguousarray(x[rs.choice(n, k, replace=False)].astype(np.float64))
    xd = torch.from_numpy(x).to(gpu)
    want = ob.som_batch(x.astype(np.float64), w, xdim, ydim, 1, (0.05, 0.01), default_radius_range(xdim, ydim), 4)
    for mode in ["fused", "plain"]:
        kern = HipKernels()
        if mode == "plain":
            class K2(HipKernels):
                update_prepare = None
            kern = HipKernels()
            kern.__class__ = type("Plain", (), {"accumulate": HipKernels.accumulate, "batch_update": HipKernels.batch_update, "__init__": HipKernels.__init__})
        tr = BatchSOMTrainer(xdim, ydim, c, gpu, batch_steps=4, kernels=kern)
        res = []
        for rep in range(3):
            wb = torch.from_numpy(w.copy()).to(gpu)
            tr.train(xd, wb, num_passes=1)
            res.append(np.abs(wb.cpu().numpy() / want - 1).max())
        print(xdim, ydim, c, dtype.__name__, mode, ["%.2e" % r for r in res], flush=True)
