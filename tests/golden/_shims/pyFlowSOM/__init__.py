"""Shim for the ABSENT third-party pyFlowSOM==0.1.16 (test-infra only).

Backed by the build's CPU oracle (oracle/pxsom_oracle.c) through tests/oracle_binding.py, with
this build's documented seed handling (ark_analysis_amd.flowsom.som_init_and_order).  It exists
so that the reference's *own Python* around the two pyFlowSOM calls can be executed here to
generate end-to-end fixtures; the SOM arithmetic inside those fixtures is oracle-of-record =
build restatement ("parity unpinned", see oracle/README.md).
"""
import os

import numpy as np


def som(data, xdim=10, ydim=10, rlen=10, alpha_range=(0.05, 0.01), radius_range=None,
        distf=2, nodes=None, importance=None, seed=None):
    from tests import oracle_binding as ob
    from ark_analysis_amd.flowsom import som_init_and_order, default_radius_range
    data = np.ascontiguousarray(data, dtype=np.float64)
    n = data.shape[0]
    init_idx, order = som_init_and_order(n, xdim * ydim, rlen, seed)
    codes = data[init_idx].copy() if nodes is None else np.array(nodes, dtype=np.float64)
    if radius_range is None:
        radius_range = default_radius_range(xdim, ydim)
    steps = os.environ.get("PXSOM_SHIM_BATCH_STEPS")
    if steps:   # make_golden.py g7b: the same reference pipeline with the build's batch rule underneath -- as the drop-in
        # classes run it on their binary64 tables: the rows join the statistics rounded to the run's quantum
        from ark_analysis_amd import _capi
        vmax = float(np.abs(data[np.isfinite(data)]).max()) if data.size else 0.0
        quantum = float(_capi.lib().pxsom_exact_sum_quantum(vmax, n // int(steps) + 1))
        return ob.som_batch_sched(data, codes, xdim, ydim, rlen, alpha_range, radius_range, int(steps), list(range(int(steps) + 1)),
                                  quantum=quantum)
    return ob.som_online(data, codes, xdim, ydim, rlen, alpha_range, radius_range, order)


def map_data_to_nodes(nodes, newdata, distf=2):
    from tests import oracle_binding as ob
    return ob.map_data_to_nodes(np.asarray(nodes, dtype=np.float64),
                                np.asarray(newdata, dtype=np.float64))
