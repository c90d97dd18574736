"""CPU-side checks of the C-ABI library and the host logic that needs no GPU."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "pxsom.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pxsom_[a-z0-9_]+)\s*\(", text)))


def test_library_loads_and_exports_every_declared_symbol():
    """No compute call: dlopen + dlsym of everything include/pxsom.h declares."""
    from ark_analysis_amd import _capi
    lib = _capi.lib()
    declared = _declared_symbols()
    assert len(declared) >= 14
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/pxsom.h but not exported"
    assert set(declared) == set(_capi.SYMBOLS), "ctypes prototype table and header disagree"
    assert lib.pxsom_abi_version() == _capi.ABI_VERSION == 9


def test_argument_validation_without_gpu():
    """Status codes + pxsom_last_error for bad arguments (rejected before any HIP call)."""
    from ark_analysis_amd import _capi
    lib = _capi.lib()
    assert lib.pxsom_assign_workspace_bytes(1000, 22, 100) > 1000 * 4
    assert lib.pxsom_assign_workspace_bytes(1000, 0, 100) == 0
    assert lib.pxsom_assign_workspace_bytes(1000, 22, 5000) == 0
    rc = lib.pxsom_assign(None, 10, 5000, 5000, 0, None, 100, None, None, None, 0, None)
    assert rc == -2 and b"c=5000" in lib.pxsom_last_error()
    assert lib.pxsom_assign_workspace_bytes(1000, 400, 100) > 400 * 100 * 8      # wide rows: the transposed codebook copy
    assert lib.pxsom_exact_sum_quantum(4.0, 200_000) == 2.0 ** (3 + 18 - 52) and lib.pxsom_exact_sum_quantum(0.0, 10) == 0.0
    rc = lib.pxsom_assign(None, 10, 22, 22, 7, None, 100, None, None, None, 0, None)
    assert rc == -2
    rc = lib.pxsom_train_online(None, 0, 22, 22, 0, None, 40, 40, 1, 0.05, 0.01, 6.0, 0.0, None, None)
    assert rc == -2 and b"grid" in lib.pxsom_last_error()
    rc = lib.pxsom_cluster_sums(None, -1, 22, 22, 0, None, 100, None, None, None)
    assert rc == -1
    with pytest.raises(_capi.PxsomError):
        _capi.check(rc, "pxsom_cluster_sums")
    # dtype codes: 0 / 1 / 2 (fp32 / fp64 / fp16) are accepted, anything else is not
    assert lib.pxsom_cluster_sums(None, 0, 22, 22, 2, None, 100, None, None, None) == -1   # null tables, dtype ok
    assert lib.pxsom_cluster_sums(None, 0, 22, 22, 3, None, 100, None, None, None) == -2
    rc = lib.pxsom_batch_update_prepare(None, 10, 10, 22, None, None, 1.0, 0.05, None, 0, None)
    assert rc == -1 and b"null" in lib.pxsom_last_error()
    rc = lib.pxsom_batch_update_prepare(None, 40, 40, 22, None, None, 1.0, 0.05, None, 0, None)
    assert rc == -2
    rc = lib.pxsom_batch_accumulate(None, 10, 22, 22, 0, None, 100, None, None, None, 0, 0, None)
    assert rc == -1
    rc = lib.pxsom_pair_histogram(None, None, 10, 0, 5, None, None)
    assert rc == -1 and b"sizes" in lib.pxsom_last_error()
    assert lib.pxsom_pair_histogram(None, None, 10, 5, 5, None, None) == -1
    assert lib.pxsom_cluster_mask_workspace_bytes(1024, 1024) == 8 * 1024 * 1024
    assert lib.pxsom_cluster_mask_workspace_bytes(0, 1024) == 0
    rc = lib.pxsom_cluster_mask(None, None, None, 10, None, 5, 0, 64, None, None, None, 0, None)
    assert rc == -1 and b"sizes" in lib.pxsom_last_error()
    rc = lib.pxsom_cluster_mask(None, None, None, 10, None, 5, 64, 64, None, None, None, 0, None)
    assert rc == -1 and b"null" in lib.pxsom_last_error()


def test_pipeline_stays_loud_without_gpu(tmp_path):
    """cluster_pixels on a CPU box must raise (no silent CPU route through the Arrow fast path either)."""
    if torch.cuda.is_available():
        pytest.skip("CPU-box behaviour")
    import pandas as pd
    from ark_analysis_amd import flowsom
    from ark_analysis_amd.fov_tables import write_dataframe
    from ark_analysis_amd.phenotyping import cluster_helpers, pixel_som_clustering
    chans = ["a", "b"]
    for d in ("pixel_mat_data", "pixel_mat_subsetted"):
        (tmp_path / d).mkdir()
        df = pd.DataFrame(np.random.rand(50, 2), columns=chans)
        df["fov"], df["row_index"], df["column_index"] = "fov0", 0, 0
        write_dataframe(df, str(tmp_path / d / "fov0.feather"))
    write_dataframe(pd.DataFrame(np.ones((1, 2)), columns=chans), str(tmp_path / "norm.feather"))
    write_dataframe(pd.DataFrame(np.random.rand(4, 2), columns=chans), str(tmp_path / "w.feather"))
    som = cluster_helpers.PixelSOMCluster(str(tmp_path / "pixel_mat_subsetted"), str(tmp_path / "norm.feather"),
                                          str(tmp_path / "w.feather"), ["fov0"], chans, xdim=2, ydim=2)
    assert som.weights is not None and flowsom.map_data_to_nodes is not None
    with pytest.raises(RuntimeError, match="no HIP device"):
        pixel_som_clustering.cluster_pixels(["fov0"], str(tmp_path), som)


def test_glibc_rand_host_helper_matches_libc():
    from ark_analysis_amd import _capi
    libc = ctypes.CDLL("libc.so.6")
    for seed in (42, 7):
        libc.srand(seed)
        ref = np.array([libc.rand() for _ in range(1000)], dtype=np.int32)
        np.testing.assert_array_equal(_capi.glibc_rand(seed, 1000), ref)


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-box behaviour")
def test_product_path_fails_loudly_without_gpu():
    """No CPU fallback: the pyFlowSOM-compatible entry points raise when no HIP device exists."""
    from ark_analysis_amd import flowsom
    x = np.random.rand(300, 4)
    with pytest.raises(RuntimeError, match="no HIP device"):
        flowsom.som(x, xdim=5, ydim=5, rlen=1, seed=1)
    with pytest.raises(RuntimeError, match="no HIP device"):
        flowsom.map_data_to_nodes(x[:25], x)
    with pytest.raises(RuntimeError, match="no HIP device"):
        flowsom.cluster_sums(x, np.ones(300, dtype=np.int32), 3)


def test_product_package_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under ark_analysis_amd/ may reference it."""
    pkg = os.path.join(ROOT, "ark_analysis_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle_binding" not in text and "libpxsom_oracle" not in text, f


def test_seed_handling_and_radius_defaults():
    from ark_analysis_amd import flowsom
    assert flowsom.default_radius_range(10, 10) == (6.0, 0.0)
    assert flowsom.default_radius_range(20, 20) == (11.0, 0.0)
    assert flowsom.default_radius_range(20, 10) == (9.0, 0.0)
    i1, o1 = flowsom.som_init_and_order(1000, 100, 2, 42)
    i2, o2 = flowsom.som_init_and_order(1000, 100, 2, 42)
    assert np.array_equal(i1, i2) and np.array_equal(o1, o2)        # same seed -> same inputs
    assert len(set(i1.tolist())) == 100 and o1.shape == (2000,) and o1.min() >= 0 and o1.max() < 1000
    i3, o3 = flowsom.som_init_and_order(1000, 100, 2, 43)
    assert not np.array_equal(o1, o3)
    with pytest.raises(ValueError):
        flowsom.som_init_and_order(50, 100, 1, 1)


def test_host_utils_natsort_and_verifiers(tmp_path):
    from ark_analysis_amd import host_utils as hu
    assert hu.natsorted(["chan10", "chan2", "chan1"]) == ["chan1", "chan2", "chan10"]
    for n in ("fov10.feather", "fov2.feather", ".hidden", "fov1.feather"):
        (tmp_path / n).write_text("x")
    (tmp_path / "sub").mkdir()
    assert hu.list_files(str(tmp_path), substrs=".feather") == ["fov1.feather", "fov2.feather", "fov10.feather"]
    assert hu.remove_file_extensions(["a.feather"]) == ["a"]
    with pytest.raises(FileNotFoundError):
        hu.validate_paths([str(tmp_path / "nope")])
    with pytest.raises(ValueError):
        hu.verify_in_list(a=["x", "y"], b=["x"])
    with pytest.raises(ValueError):
        hu.verify_same_elements(enforce_order=True, a=["x", "y"], b=["y", "x"])
    assert hu.verify_same_elements(a=["x", "y"], b=["y", "x"])


def test_bench_refuses_more_gpus_than_the_box_has():
    """`bench.py --gpus N` from a plain shell spawns its own ranks -- and must not run a smaller job under that
    name when the box has fewer devices (here: none)."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("box has two devices")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert "--gpus 2" in r.stderr and "device" in r.stderr


def test_bench_picks_the_launch_over_all_rows_from_counter_rows():
    """bench.py's PMC post-processing: training-step launches of the same kernel (same capped grid, small counters)
    must not dilute the figures of the launch over all rows; repeats of that launch are averaged."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("pxsom_bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rows = []
    for did in range(64):                                   # 64 training-step launches
        rows.append({"Dispatch_Id": str(did), "Counter_Name": "FETCH_SIZE", "Counter_Value": "1500.5"})
    for did, v in ((100, 450000.0), (101, 451000.0)):       # two timed launches over all rows, split in two rows each
        rows.append({"Dispatch_Id": str(did), "Counter_Name": "FETCH_SIZE", "Counter_Value": str(v / 2)})
        rows.append({"Dispatch_Id": str(did), "Counter_Name": "FETCH_SIZE", "Counter_Value": str(v / 2)})
    assert bench.select_launch_counters(rows, ["FETCH_SIZE"]) == {"FETCH_SIZE": 450500.0}
    multi = [{"Dispatch_Id": "1", "Counter_Name": "GRBM_GUI_ACTIVE", "Counter_Value": "10"},
             {"Dispatch_Id": "1", "Counter_Name": "SQ_INSTS_MFMA", "Counter_Value": "3"},
             {"Dispatch_Id": "2", "Counter_Name": "GRBM_GUI_ACTIVE", "Counter_Value": "4000"},
             {"Dispatch_Id": "2", "Counter_Name": "SQ_INSTS_MFMA", "Counter_Value": "900"}]
    assert bench.select_launch_counters(multi, ["SQ_INSTS_MFMA", "GRBM_GUI_ACTIVE"]) == {"SQ_INSTS_MFMA": 900.0,
                                                                                       "GRBM_GUI_ACTIVE": 4000.0}
    assert bench.select_launch_counters([], ["FETCH_SIZE"]) == {}


def test_binary16_fixed_point_conversion_is_exact_for_every_finite_half():
    """The mean-table kernel turns a binary16 value into 2^-24 fixed point as bits(v + 1.5 * 2^28) - bits(1.5 * 2^28)
    (csrc/pxsom_train.hip, cluster_sums_kernel): checked here for all 63 488 finite bit patterns, signed zeros and
    subnormals included; and the vector-wide Inf / NaN test ((bits & 0x7fff) + 0x0400 reaches bit 15) for all 65 536."""
    bits = np.arange(1 << 16, dtype=np.uint16)
    halves = bits.view(np.float16)
    finite = np.isfinite(halves)
    v = halves[finite].astype(np.float64)
    magic = np.float64(1.5 * 2.0 ** 28)
    assert magic.view(np.uint64) == np.uint64(0x41B8000000000000)
    shifted = v + magic
    q = (shifted.view(np.uint64) - magic.view(np.uint64)).view(np.int64)      # wraps for negative values, as the kernel's
    np.testing.assert_array_equal(q, np.round(v * 2.0 ** 24).astype(np.int64))
    np.testing.assert_array_equal(q.astype(np.float64) * 2.0 ** -24, v)       # exact both ways
    special = (((bits & 0x7FFF).astype(np.uint32) + 0x0400) & 0x8000) != 0
    np.testing.assert_array_equal(special, ~finite)


def test_screening_rule_of_the_long_list_exact_kernel_never_drops_a_possible_winner():
    """Numpy model of bmu_exact_screened_kernel's screening rule (csrc/pxsom_assign.hip): a node stays a candidate
    unless its binary32 squared distance exceeds T = ((D_ref (1 + 1e-12) + delta)^2 (1 + eta), delta =
    2^-24 (|x| + max|w|) 1.001 + 1e-30, eta = (C + 4) 2^-24.  Whatever order the binary32 sum is taken in, every node
    whose binary64 (oracle-order) distance is <= D_ref must pass -- checked on crowded codebooks, rows sitting on
    nodes, tiny and large magnitudes, binary64 rows (rounded to binary32 for the screening pass)."""
    rs = np.random.RandomState(20260928)
    u24 = 2.0 ** -24
    for case in range(300):
        c = int(rs.choice([1, 2, 7, 22, 40, 100, 128]))
        k = int(rs.choice([4, 100, 400]))
        scale = 10.0 ** rs.uniform(-6, 6)
        w = rs.rand(k, c) * scale
        w[k // 2:] = w[: k - k // 2] * (1.0 + 10.0 ** rs.uniform(-12, -3))        # crowded pairs
        x = w[rs.randint(0, k)] * (1.0 + 10.0 ** rs.uniform(-9, -2) * rs.standard_normal(c))
        if case % 5 == 0:
            x = w[rs.randint(0, k)].copy()                                        # distance 0 to a node
        if case % 2 == 0:
            x = x.astype(np.float32).astype(np.float64)                           # binary32 storage
        dist = np.empty(k)
        for node in range(k):                                                     # the oracle's order
            acc = 0.0
            for j in range(c):
                t = x[j] - w[node, j]
                acc += t * t
            dist[node] = np.sqrt(acc)
        ref = int(np.argsort(dist)[min(k - 1, rs.randint(0, 3))])                 # the filter's proposal: a near-best node
        x32, w32 = x.astype(np.float32), w.astype(np.float32)
        n2 = np.float32(np.sum(x32 * x32, dtype=np.float32))
        wn = float(np.sqrt((w * w).sum(axis=1)).max()) * (1.0 + 1e-6)
        delta = u24 * (np.sqrt(float(n2)) + wn) * 1.001 + 1.0e-30
        eta = (c + 4) * u24
        tt = dist[ref] * (1.0 + 1.0e-12) + delta
        thr = np.float32(max(float(np.float32(tt * tt * (1.0 + eta) * (1.0 + 2.0 ** -20))), 1.2e-38))
        t32 = x32[None, :] - w32
        sq = t32 * t32
        orders = [np.sum(sq, axis=1, dtype=np.float32),                                            # numpy's pairwise order
                  np.add.accumulate(sq, axis=1, dtype=np.float32)[:, -1],                          # sequential
                  np.add.accumulate(sq[:, ::-1], axis=1, dtype=np.float32)[:, -1],                 # reversed
                  sum(np.add.accumulate(sq[:, r::8], axis=1, dtype=np.float32)[:, -1] for r in range(min(8, c)))]   # 8 chains
        must_pass = dist <= dist[ref]
        for d32 in orders:
            assert not (d32[must_pass] > thr).any(), (case, c, k, scale)
