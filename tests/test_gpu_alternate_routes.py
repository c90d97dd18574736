"""Kernel routes that only some shapes take, each checked against the oracle in a process of its own: the two-tile one-pass
kernel (binary64 rows: labels only, labels + tables), the packed-K filter (binary16 rows of wide codebooks).
(Until round 5 this file toggled environment switches of libpxsom -- PXSOM_ONEPASS, PXSOM_PACKED_TWO, ...  Round 6: the library
reads no environment variable any more; the opt-in kernels behind those switches were measured slower and removed, what is left
here are routes a shape takes BY DEFAULT.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHECK = r"""
import sys
import numpy as np, torch
sys.path.insert(0, %(root)r)
from ark_analysis_amd import som_device as sd, synth
from tests import oracle_binding as ob
dev = torch.device("cuda:0")
rs = np.random.RandomState(3)
for dtype, n, c, k in %(cases)s:
    x = synth.make_fov_numpy(n, c, seed=11, dtype=np.float64)
    x[rs.randint(0, n, 40)] = x[rs.randint(0, n, 40)]                       # duplicated rows
    xt = torch.from_numpy(x).to(getattr(torch, dtype)).to(dev)
    host = xt.cpu().double().numpy()
    w = host[rs.choice(n, k, replace=n < k)].copy()
    w[k - 1] = w[2]                                                         # a duplicate node: the first index wins
    w[5] = w[6] * (1 + 1e-9)                                                # a near tie
    wd = torch.from_numpy(w).to(dev)
    want, want_d = ob.map_data_to_nodes(w, host)
    lab, dist = sd.assign(xt, wd, want_dists=True)
    assert np.array_equal(lab.cpu().numpy(), want), (dtype, n, c, k, "assign")
    assert np.array_equal(dist.cpu().numpy(), want_d), (dtype, n, c, k, "distances")
    lab2, sums, counts = sd.assign_sums(xt, wd)
    ws, wc = ob.cluster_sums(host, want, k)
    assert np.array_equal(lab2.cpu().numpy(), want), (dtype, n, c, k, "assign_sums labels")
    assert np.array_equal(counts.cpu().numpy(), wc), (dtype, n, c, k, "counts")
    np.testing.assert_allclose(sums.cpu().numpy(), ws, rtol=1e-9, atol=1e-6 * float(np.abs(w).max()))
print("ok")
"""


@pytest.mark.parametrize("route,cases", [
    ("two-tile kernel, binary64 rows", [("float64", 20_011, 22, 100), ("float64", 5_000, 32, 100), ("float64", 64, 2, 100), ("float64", 12_345, 8, 98)]),
    ("packed-K filter, binary16 rows", [("float16", 30_000, 40, 400), ("float16", 7_001, 64, 256)]),
])
def test_route_matches_the_oracle(route, cases):
    res = subprocess.run([sys.executable, "-c", CHECK % {"root": ROOT, "cases": repr(cases)}], env=dict(os.environ), capture_output=True,
                         text=True, timeout=900)
    assert res.returncode == 0 and res.stdout.strip().endswith("ok"), route + ": " + res.stdout[-2000:] + res.stderr[-4000:]
