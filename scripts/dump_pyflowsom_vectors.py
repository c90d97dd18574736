#!/usr/bin/env python
"""Dump golden vectors from the REAL pyFlowSOM (the package ark pins: pyFlowSOM==0.1.16, uv.lock:3005-3013).

pyFlowSOM is absent from the build image and from /root/reference, so the SOM kernels' oracle
(oracle/pxsom_oracle.c orc_som_online / orc_map_data_to_nodes) is a restatement from recollection: "parity
unpinned".  This one-file script is the route to "pinned": run it anywhere pyFlowSOM is importable

    pip install pyFlowSOM==0.1.16 numpy
    python scripts/dump_pyflowsom_vectors.py            # writes tests/golden/g11_pyflowsom.npz

commit the file, and tests/test_pyflowsom_vectors.py (CPU: oracle; -m gpu: the HIP path) compares against it --
including the seed -> (initial nodes, presentation order) mapping of ark_analysis_amd.flowsom.som, which today is a
documented guess.  It needs nothing from this repository but numpy; the arrays it stores are inputs and outputs only.

Cases: C in {4, 22, 40} x K in {100 (10x10), 400 (20x20)} x rlen in {1, 2}, seeds 42 / 7; plus, where this
pyFlowSOM accepts ``nodes=``, the same runs with explicit initial nodes (separates the init rule from the order
stream), an exact-tie pair and a NaN row for map_data_to_nodes; plus two multi-pass runs on data clipped to [0, 0.9]
(every |x - w| < 1: tells FlowSOM's `change += abs(tmp)` -- integer abs -- from fabs, see flowsom.RECALLED["change_abs"]).
"""
import os
import sys

import numpy as np


def main():
    try:
        import pyFlowSOM
        from pyFlowSOM import map_data_to_nodes, som
    except Exception as err:   # noqa: BLE001
        sys.exit("pyFlowSOM is not importable here (%s): run this where `pip install pyFlowSOM==0.1.16` works" % err)
    out = {"pyflowsom_version": np.array(getattr(pyFlowSOM, "__version__", "unknown")),
           "numpy_version": np.array(np.__version__)}
    cases = []
    for c in (4, 22, 40):
        for (xdim, ydim) in ((10, 10), (20, 20)):
            for rlen in (1, 2):
                cases.append((c, xdim, ydim, rlen, 42 if rlen == 1 else 7, False))
    # two-pass runs on data clipped to [0, 0.9]: every |x - w| stays below 1, so the two readings of the early-stop
    # accumulator (fabs / integer abs: flowsom.RECALLED["change_abs"], oracle ORC_V_INT_ABS) give different codebooks
    cases.append((22, 10, 10, 2, 11, True))
    cases.append((8, 10, 10, 3, 12, True))
    for idx, (c, xdim, ydim, rlen, seed, clipped) in enumerate(cases):
        tag = "case%02d" % idx
        rs = np.random.RandomState(1000 + idx)
        n = 1500 if xdim == 10 else 2400
        x = rs.gamma(0.7, 0.4, size=(n, c))
        x[rs.rand(n, c) < 0.2] = 0.0
        if clipped:
            x = np.minimum(x, 0.9)
        x = np.ascontiguousarray(x, dtype=np.float64)
        codes = som(x, xdim=xdim, ydim=ydim, rlen=rlen, alpha_range=(0.05, 0.01), seed=seed)
        codes = np.asarray(codes, dtype=np.float64).reshape(xdim * ydim, -1)
        codes_again = np.asarray(som(x, xdim=xdim, ydim=ydim, rlen=rlen, alpha_range=(0.05, 0.01), seed=seed),
                                 dtype=np.float64).reshape(xdim * ydim, -1)
        test = np.concatenate([x[:600], codes[[0, 3]], 0.5 * (codes[3:4] + codes[4:5])])
        tie_codes = codes.copy()
        tie_codes[-1] = tie_codes[len(tie_codes) // 2]              # an exact tie pair: which label comes back?
        labels, dists = map_data_to_nodes(codes, test)
        tlabels, tdists = map_data_to_nodes(tie_codes, test)
        out.update({tag + "_grid": np.array([xdim, ydim, rlen, seed], dtype=np.int64), tag + "_x": x,
                    tag + "_codes": codes, tag + "_same_seed_equal": np.array(np.array_equal(codes, codes_again)),
                    tag + "_test": test, tag + "_labels": np.asarray(labels).astype(np.int64),
                    tag + "_dists": np.asarray(dists, dtype=np.float64),
                    tag + "_tie_labels": np.asarray(tlabels).astype(np.int64)})
        try:   # explicit initial nodes, if this version takes them
            init = x[np.random.RandomState(seed).choice(n, xdim * ydim, replace=False)].copy()
            cn = som(x, xdim=xdim, ydim=ydim, rlen=rlen, alpha_range=(0.05, 0.01), nodes=init.copy(), seed=seed)
            out[tag + "_init"] = init
            out[tag + "_codes_from_init"] = np.asarray(cn, dtype=np.float64).reshape(xdim * ydim, -1)
        except Exception:   # noqa: BLE001
            pass
        try:
            nan_row = np.full((1, c), np.nan)
            nl, nd = map_data_to_nodes(codes, nan_row)
            out[tag + "_nan_label"] = np.asarray(nl).astype(np.int64)
        except Exception:   # noqa: BLE001
            pass
    out["n_cases"] = np.array(len(cases), dtype=np.int64)
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.path.join(os.path.dirname(here), "tests", "golden", "g11_pyflowsom.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "with", len(cases), "cases from pyFlowSOM", out["pyflowsom_version"])


if __name__ == "__main__":
    main()
