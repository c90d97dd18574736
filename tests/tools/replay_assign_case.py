"""Replays one case of tests/test_gpu_fuzz_parity.py::test_fuzz_assign_and_sums (test tooling: uses the oracle).
python -m tests.tools.replay_assign_case <case index>"""
import sys
import numpy as np
import torch
from tests import oracle_binding as oracle
from tests.test_gpu_fuzz_parity import SEED, _case, _codebook
from ark_analysis_amd import som_device

target = int(sys.argv[1])
rs = np.random.RandomState(SEED)
for case in range(target + 1):
    x, host, xdim, ydim, kind, dtype = _case(rs)
    k = xdim * ydim
    w = _codebook(rs, host, k, kind)
n, c = host.shape
print("case", target, "n", n, "c", c, "k", k, dtype, kind, "ldx", x.stride(0))
wd = torch.from_numpy(w).cuda()
want, dist = oracle.map_data_to_nodes(w, host)
labels, _ = som_device.assign(x, wd)
print("two passes: labels differing from the oracle:", int((labels.cpu().numpy() != want).sum()),
      "listed rows:", som_device.last_exact_rows(som_device.assign.last_workspace))
for rep in range(3):
    lab2, s2, c2 = som_device.assign_sums(x, wd)
    got = lab2.cpu().numpy()
    bad = np.flatnonzero(got != want)
    print("one pass, run %d: %d labels differ; rows %s got %s want %s" % (rep, len(bad), bad[:8].tolist(), got[bad[:8]].tolist(), want[bad[:8]].tolist()))
    if len(bad):
        r = bad[0]
        d = np.sqrt(((host[r][None, :] - w) ** 2).sum(1))
        order = np.argsort(d)[:3]
        print("   row %d: nearest nodes %s distances %s; groups of 64: row %% 64 = %d, group %d of %d" % (r, (order + 1).tolist(), d[order].tolist(), r % 64, r // 64, (n + 63) // 64))
    ws, wc = oracle.cluster_sums(host, want, k)
    print("   counts equal:", bool(np.array_equal(c2.cpu().numpy(), wc)), " total counted:", int(c2.sum().item()), "of", n)
