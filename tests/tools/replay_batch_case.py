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
    m = int(rs.choice([1, 2, 4, 8, 16]))
    passes = int(rs.choice([1, 1, 2]))
    w0 = _codebook(rs, host, k, kind)
print("case", target, "n", n, "c", c, "grid", xdim, ydim, dtype, kind, "m", m, "passes", passes, "ldx", x.stride(0))
alpha, radius = (0.05, 0.01), default_radius_range(xdim, ydim)
print("radius", radius)
total = m * passes
unfused = len(sys.argv) > 2 and sys.argv[2] == "unfused"
st = som_device.BatchTrainState(n, c, xdim, ydim, m, x.device)
st.wbuf[0].copy_(torch.from_numpy(w0))
prev = None
for g in range(total + 1):
    if g < total:
        som_device.batch_train_steps(x, st, g, g + 1, total, alpha, radius, unfused=unfused)
        w_g = st.wbuf[g % 2].cpu().numpy().reshape(k, c)
    else:
        out = torch.empty((k, c), dtype=torch.float64, device=x.device)
        som_device.batch_train_finish(st, total, total, alpha, radius, out)
        w_g = out.cpu().numpy()
    if prev is not None:
        want_w = oracle.batch_update(prev[0], xdim, ydim, prev[1], prev[2], prev[3], prev[4])
        rel = np.abs(w_g - want_w) / np.maximum(np.abs(want_w), 1e-300)
        node, ch = np.unravel_index(np.argmax(rel), rel.shape)
        print("update %d (thr %.3g alpha %.4g): max rel err %.3g at node %d channel %d: gpu %.17g oracle %.17g old w %.17g" % (
            g - 1, prev[3], prev[4], rel.max(), node, ch, w_g[node, ch], want_w[node, ch], prev[0][node, ch]))
        r = int(np.floor(prev[3])); nx, ny = node // ydim, node % ydim
        win = [bx * ydim + by for bx in range(max(0, nx - r), min(xdim - 1, nx + r) + 1) for by in range(max(0, ny - r), min(ydim - 1, ny + r) + 1)]
        den = prev[2][win].sum(); num = prev[1][win, ch].sum()
        print("   window of %d nodes, den %d, mean %.17g, mean - w %.3g, gain %.17g" % (len(win), den, num / max(den, 1), num / max(den, 1) - prev[0][node, ch], -np.expm1(den * np.log1p(-prev[4]))))
    if g == total:
        break
    rows = host[g % m::m]
    ring = st.ring[g % 3].cpu().numpy()
    got_s, got_c = ring[:k * c].reshape(k, c), ring[k * c:]
    thr = radius[0] - (radius[0] - radius[1]) * g / total
    a = alpha[0] - (alpha[0] - alpha[1]) * g / total
    prev = (w_g, got_s.copy(), got_c.astype(np.int64), 0.5 if thr < 1.0 else thr, a)
sys.exit(0)
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
