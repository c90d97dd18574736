"""Replays one case of tests/test_gpu_fuzz_parity.py::test_fuzz_batch_training step by step (test tooling: uses
the oracle).  python -m tests.tools.replay_batch_case <case index>"""
import sys
import numpy as np
import torch
from tests import oracle_binding as oracle
from tests.test_gpu_fuzz_parity import SEED, _case, _codebook
from ark_analysis_amd import som_device
from ark_analysis_amd.flowsom import default_radius_range

target = int(sys.argv[1])
rs = np.random.RandomState(SEED + 1)
for case in range(target + 1):
    x, host, xdim, ydim, kind, dtype = _case(rs, max_work=1.5e7)
    n, c = host.shape
    k = xdim * ydim
    if kind == "quantised" or n < 2:
        continue
    m = int(rs.choice([1, 2, 4, 8, 16]))
    passes = int(rs.choice([1, 1, 2]))
    w0 = _codebook(rs, host, k, kind)
print("case", target, "n", n, "c", c, "grid", xdim, ydim, dtype, kind, "m", m, "passes", passes, "ldx", x.stride(0))
alpha, radius = (0.05, 0.01), default_radius_range(xdim, ydim)
print("radius", radius)
total = m * passes
for steps in range(1, total + 1):
    # oracle after `steps` steps of `total`: run the loop by hand
    w = w0.copy()
    for g in range(steps):
        rows = host[g % m::m]
        lab, _ = oracle.map_data_to_nodes(w, rows)
        s, cnt = oracle.cluster_sums(rows, lab, k)
        thr = radius[0] - (radius[0] - radius[1]) * g / total
        thr = 0.5 if thr < 1.0 else thr
        a = alpha[0] - (alpha[0] - alpha[1]) * g / total
        w = oracle.batch_update(w, xdim, ydim, s, cnt, thr, a)
    st = som_device.BatchTrainState(n, c, xdim, ydim, m, x.device)
    st.wbuf[0].copy_(torch.from_numpy(w0))
    som_device.batch_train_steps(x, st, 0, steps, total, alpha, radius)
    out = torch.empty((k, c), dtype=torch.float64, device=x.device)
    som_device.batch_train_finish(st, steps, total, alpha, radius, out)
    got = out.cpu().numpy()
    ring = st.ring[(steps - 1) % 3].cpu().numpy()
    ds = np.abs(ring[:k * c].reshape(k, c) - s).max()
    dc = np.abs(ring[k * c:] - cnt).max()
    err = np.abs(got - w).max() / np.abs(w).max()
    print("steps %2d: last stats |dsum| %.3g |dcount| %.3g; codebook max err (rel to max) %.3g" % (steps, ds, dc, err))
    if err > 1e-9:
        bad = np.argwhere(np.abs(got - w) > 1e-9 * np.abs(w).max())
        print("  first bad entries (node, channel):", bad[:6].tolist(), "nodes affected:", len(set(bad[:, 0].tolist())))
        break
