"""Consumes tests/golden/g11_pyflowsom.npz -- vectors dumped from the REAL pyFlowSOM 0.1.16 by
scripts/dump_pyflowsom_vectors.py -- when that file exists; skipped otherwise (pyFlowSOM is absent from the build
image: SOM parity is "unpinned" until somebody runs the dump script where the package installs and commits the
file).  CPU: the oracle; -m gpu: the HIP path through ark_analysis_amd.flowsom.

map_data_to_nodes needs no recollection beyond the arithmetic (labels + distances must be equal, bit for bit).
som() additionally depends on the recalled seed -> (initial nodes, order) mapping and on a handful of recalled
details; on a mismatch the test sweeps the named switches (oracle ORC_V_*, flowsom.RECALLED) and reports which
combination reproduces pyFlowSOM, so the failure says what to change."""
import itertools
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g11_pyflowsom.npz")
needs_vectors = pytest.mark.skipif(not os.path.exists(GOLD), reason="tests/golden/g11_pyflowsom.npz not present "
                                   "(run scripts/dump_pyflowsom_vectors.py where pyFlowSOM==0.1.16 installs)")


def _cases():
    g = np.load(GOLD)
    for i in range(int(g["n_cases"])):
        yield "case%02d" % i, g


@needs_vectors
def test_oracle_bmu_search_equals_pyflowsom(oracle):
    for tag, g in _cases():
        labels, dists = oracle.map_data_to_nodes(g[tag + "_codes"], g[tag + "_test"])
        np.testing.assert_array_equal(labels, g[tag + "_labels"], err_msg=tag)
        np.testing.assert_array_equal(dists, g[tag + "_dists"], err_msg=tag)
        codes = g[tag + "_codes"].copy()
        codes[-1] = codes[len(codes) // 2]
        tl, _ = oracle.map_data_to_nodes(codes, g[tag + "_test"])
        if not np.array_equal(tl, g[tag + "_tie_labels"]):
            alt, _ = oracle.map_data_to_nodes_variant(codes, g[tag + "_test"], oracle.V_LAST_MINIMUM)
            pytest.fail("%s: tie-break differs from pyFlowSOM; ORC_V_LAST_MINIMUM matches: %s"
                        % (tag, np.array_equal(alt, g[tag + "_tie_labels"])))
        if tag + "_nan_label" in g.files:
            nl, _ = oracle.map_data_to_nodes(g[tag + "_codes"], np.full((1, g[tag + "_x"].shape[1]), np.nan))
            np.testing.assert_array_equal(nl, g[tag + "_nan_label"])


def _sweep(oracle, g, tag, init, want):
    """Which combination of the recalled-detail switches reproduces ``want`` from ``init``?"""
    from ark_analysis_amd import flowsom
    xdim, ydim, rlen, seed = (int(v) for v in g[tag + "_grid"])
    x = g[tag + "_x"]
    hits = []
    for stream, quant, node_order, variant in itertools.product(
            ("glibc_rand", "numpy_randint", "numpy_sample"), (0.67, 0.5, 0.75), ("xy", "yx"), range(32)):
        _, order = flowsom.som_init_and_order(len(x), xdim * ydim, rlen, seed, order_stream=stream)
        rr = flowsom.default_radius_range(xdim, ydim, quantile=quant)
        got = oracle.som_online(x, init, xdim, ydim, rlen, (0.05, 0.01), rr, order, variant=variant, node_order=node_order)
        if np.array_equal(got, want):
            hits.append(dict(order_stream=stream, radius_quantile=quant, node_order=node_order, oracle_variant=variant))
    return hits


@needs_vectors
def test_oracle_training_equals_pyflowsom(oracle):
    from ark_analysis_amd import flowsom
    for tag, g in _cases():
        assert bool(g[tag + "_same_seed_equal"]), "pyFlowSOM itself is not reproducible for one seed?"
        xdim, ydim, rlen, seed = (int(v) for v in g[tag + "_grid"])
        x = g[tag + "_x"]
        init_idx, order = flowsom.som_init_and_order(len(x), xdim * ydim, rlen, seed)
        rr = flowsom.default_radius_range(xdim, ydim)
        if tag + "_init" in g.files:     # explicit initial nodes: isolates the order stream + the loop
            got = oracle.som_online(x, g[tag + "_init"], xdim, ydim, rlen, (0.05, 0.01), rr, order)
            if not np.array_equal(got, g[tag + "_codes_from_init"]):
                pytest.fail("%s: training from explicit nodes differs from pyFlowSOM; switch combinations that "
                            "match: %s" % (tag, _sweep(oracle, g, tag, g[tag + "_init"], g[tag + "_codes_from_init"])))
        got = oracle.som_online(x, x[init_idx], xdim, ydim, rlen, (0.05, 0.01), rr, order)
        if not np.array_equal(got, g[tag + "_codes"]):
            pytest.fail("%s: som(seed=%d) differs from pyFlowSOM (initial-node rule and / or loop details); switch "
                        "combinations that match with the numpy_choice init: %s"
                        % (tag, seed, _sweep(oracle, g, tag, x[init_idx], g[tag + "_codes"])))


@needs_vectors
@pytest.mark.gpu
def test_hip_path_equals_pyflowsom(gpu):
    from ark_analysis_amd import flowsom
    for tag, g in _cases():
        xdim, ydim, rlen, seed = (int(v) for v in g[tag + "_grid"])
        labels, dists = flowsom.map_data_to_nodes(g[tag + "_codes"], g[tag + "_test"])
        np.testing.assert_array_equal(labels, g[tag + "_labels"], err_msg=tag)
        np.testing.assert_array_equal(dists, g[tag + "_dists"], err_msg=tag)
        codes = flowsom.som(g[tag + "_x"], xdim=xdim, ydim=ydim, rlen=rlen, alpha_range=(0.05, 0.01), seed=seed)
        np.testing.assert_array_equal(np.asarray(codes).reshape(xdim * ydim, -1), g[tag + "_codes"], err_msg=tag)


def test_recalled_switches_are_wired(oracle):
    """Without real vectors: the default of every named switch is the arithmetic the other tests pin, and each
    alternative really changes the result (so a sweep can discriminate)."""
    from ark_analysis_amd import flowsom
    rs = np.random.RandomState(0)
    x = rs.gamma(0.7, 0.4, size=(300, 5))
    init_idx, order = flowsom.som_init_and_order(300, 25, 2, 9)
    rr = flowsom.default_radius_range(5, 5)
    base = oracle.som_online(x, x[init_idx], 5, 5, 2, (0.05, 0.01), rr, order)
    np.testing.assert_array_equal(base, oracle.som_online(x, x[init_idx], 5, 5, 2, (0.05, 0.01), rr, order, variant=0, node_order="xy"))
    # the 0.5 pin is unobservable while the threshold stays in [0, 1): grid distances are integers
    np.testing.assert_array_equal(base, oracle.som_online(x, x[init_idx], 5, 5, 2, (0.05, 0.01), rr, order,
                                                          variant=oracle.V_NO_THRESHOLD_PIN))
    # ... and observable once the schedule ends below zero (then no node, not even the BMU, is inside it)
    neg = (3.0, -1.0)
    assert not np.array_equal(oracle.som_online(x, x[init_idx], 5, 5, 2, (0.05, 0.01), neg, order),
                              oracle.som_online(x, x[init_idx], 5, 5, 2, (0.05, 0.01), neg, order, variant=oracle.V_NO_THRESHOLD_PIN))
    # node numbering matters on a non-square grid
    w46 = x[init_idx[:24]]
    assert not np.array_equal(oracle.som_online(x, w46, 4, 6, 1, (0.05, 0.01), (3.0, 0.0), order[:300]),
                              oracle.som_online(x, w46, 4, 6, 1, (0.05, 0.01), (3.0, 0.0), order[:300], node_order="yx"))
    for stream in ("numpy_randint", "numpy_sample"):
        _, other = flowsom.som_init_and_order(300, 25, 2, 9, order_stream=stream)
        assert other.shape == order.shape and not np.array_equal(other, order)
        assert other.min() >= 0 and other.max() < 300
    assert flowsom.default_radius_range(10, 10) == (6.0, 0.0)
    assert flowsom.default_radius_range(10, 10, quantile=0.5)[0] < 6.0
    w = base.copy()
    w[-1] = w[3]
    first, _ = oracle.map_data_to_nodes(w, w[3:4])
    last, _ = oracle.map_data_to_nodes_variant(w, w[3:4], oracle.V_LAST_MINIMUM)
    assert first[0] == 4 and last[0] == 25
    # the two readings of the early-stop accumulator part ways on data whose differences all stay below 1: the integer
    # abs() adds nothing, the run stops at the start of its second pass (one more step runs, as in FlowSOM's loop)
    small = np.minimum(x, 0.9)
    b2 = oracle.som_online(small, small[init_idx], 5, 5, 2, (0.05, 0.01), rr, order)
    i2 = oracle.som_online(small, small[init_idx], 5, 5, 2, (0.05, 0.01), rr, order, variant=oracle.V_INT_ABS)
    assert not np.array_equal(b2, i2)
    np.testing.assert_array_equal(i2, oracle.som_online(small, small[init_idx], 5, 5, 2, (0.05, 0.01), rr, order,
                                                        variant=oracle.V_INT_ABS | oracle.V_NO_THRESHOLD_PIN))
    np.testing.assert_array_equal(b2, oracle.som_online(small, small[init_idx], 5, 5, 2, (0.05, 0.01), rr, order,
                                                        variant=oracle.V_INT_ABS | oracle.V_NO_EARLY_STOP))
    assert flowsom.RECALLED["change_abs"][0] == "fabs"
    sq, dsq = oracle.map_data_to_nodes_variant(base, x, oracle.V_COMPARE_SQUARED)
    lab, d = oracle.map_data_to_nodes(base, x)
    np.testing.assert_array_equal(sq, lab)
    np.testing.assert_array_equal(dsq, d)
