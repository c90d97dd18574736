#!/usr/bin/env python
"""bench.py -- Pixie pixel-SOM train + assign throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--config cfg2|cfg3|cfg4|cfg5]
    (N > 1 without RANK in the environment: re-executes itself under torch.distributed.run, one rank per GPU;
     fails loudly when the box has fewer than N devices)

One "step" = one pass of the hot path over one batch of synthetic input, resident in HBM before
the clock starts ("normalised pixel matrix resident" -> "labels + codebook + mean table
resident"):  batch-mode SOM training (1 pass over the training subset, `--batch-steps` mini-batch
steps, statistics all-reduced over RCCL when N > 1)  +  BMU assignment of every row  +  the per-cluster
mean-expression table over all rows (sums/counts all-reduced once when N > 1) -- labels and table from ONE pass over x
(pxsom_assign_sums; `--two-pass`: pxsom_assign + pxsom_cluster_sums, the round-1/2 form).
Workloads (`--config`; BASELINE.json configs):
  cfg2  10 FOVs 1024x1024x22 fp32 per GPU, 10x10 SOM     (configs[1] at N = 1; the metric's configuration; default)
  cfg3  25 FOVs 1024x1024x22 fp32 per GPU, 10x10 SOM     (configs[2]: 200 FOVs over 8 GPUs)
  cfg4  1e6 cells x 100 features fp32 in total, 10x10 SOM, trained on every row (configs[3]; strong scaling)
  cfg5  FOVs 2048x2048x40 fp16, 20x20 SOM                (configs[4]; `--fovs-per-gpu`, default 4)
Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline       dominant kernel = the BMU filter kernel over all rows; achieved = (C*s + 4) B/row (row read once +
                 int32 label written, DESIGN.md "K7") * rows / its HIP-event duration; traffic = HBM bytes per launch
                 from rocprofv3 PMC passes of this same script run inside the call (FETCH_SIZE x 2 + WRITE_SIZE, the
                 gfx950 correction of MI355X_MICROARCH.md), or the recorded value if the profiler cannot run.
  roofline_step  the whole step against the HBM roofline: SURVEY 8(d)'s B = C*s*(1 + f*p) + 4 bytes per row.
  mfma_util      matrix-pipe busy cycles / (SIMDs x kernel cycles) of the filter kernel, same PMC run.
  cpu_baseline   the oracle (port of the reference algorithm: online FlowSOM training on the same
                 training subset + reference-shaped BMU search) on ONE host core; bounded sample.
  batch_train    the timed training mode's codebook against orc_som_batch at FULL size (rtol 1e-9).
  online_train   the exact-online (reference-order) training kernel on the same subset, checked
                 against the oracle's codebook from the cpu_baseline leg (full-size parity).
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from ark_analysis_amd import _capi, som_device, synth  # noqa: E402
from ark_analysis_amd.distributed import (BatchSOMTrainer, all_reduce_, allreduce_cluster_tables,  # noqa: E402
                                          broadcast_codebook)
from ark_analysis_amd.flowsom import default_radius_range  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense fp16 / bf16 MFMA peak (same guide)
N_SIMDS = 1024                 # 256 CUs x 4
N_XCDS = 8                     # GRBM_GUI_ACTIVE comes back summed over the 8 XCDs

CONFIGS = {
    "cfg2": dict(kind="pixel", c=22, xdim=10, ydim=10, dtype="f32", unit_rows=1024 * 1024, units=10, frac=0.1,
                 scaling="weak", desc="{u} FOVs 1024x1024x22ch fp32 per GPU, 10x10 SOM (BASELINE.json configs[1] at N=1)"),
    "cfg3": dict(kind="pixel", c=22, xdim=10, ydim=10, dtype="f32", unit_rows=1024 * 1024, units=25, frac=0.1,
                 scaling="weak", desc="{u} FOVs 1024x1024x22ch fp32 per GPU, 10x10 SOM (BASELINE.json configs[2]: 200 FOVs over 8 GPUs)"),
    "cfg4": dict(kind="cell", c=100, xdim=10, ydim=10, dtype="f32", unit_rows=1_000_000, units=1, frac=1.0,
                 scaling="strong", desc="1e6 cells x 100 pixel-cluster-count features fp32 in total, 10x10 SOM, trained on all rows "
                                        "(BASELINE.json configs[3])"),
    "cfg5": dict(kind="pixel", c=40, xdim=20, ydim=20, dtype="f16", unit_rows=2048 * 2048, units=62, frac=0.1,
                 scaling="weak", desc="{u} FOVs 2048x2048x40ch fp16 per GPU, 20x20 SOM + consensus meta-clustering "
                                      "(BASELINE.json configs[4]: 500 FOVs over 8 GPUs = 62 per GPU, 20.8 GB of rows)"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=["cfg1"] + sorted(CONFIGS), default="cfg2",
                    help="cfg1: BASELINE.json configs[0] -- one 512x512x8 FOV through train_pixel_som -> cluster_pixels -> "
                         "generate_som_avg_files (the Python plumbing, HIP path beside the oracle-backed CPU path; one GPU)")
    ap.add_argument("--fovs-per-gpu", type=int, default=None)
    ap.add_argument("--batch-steps", default="two-phase",
                    help="training schedule: 'two-phase' (default: 6 large steps while the radius is >= 1, 16 in "
                         "the BMU-only tail) or an integer = that many equal mini-batch steps per pass (64: rounds 1-2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-online", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 counter passes (traffic / mfma_util)")
    ap.add_argument("--two-pass", action="store_true",
                    help="labels and mean table from two kernels (pxsom_assign + pxsom_cluster_sums, x read twice: 0.49 ms on "
                         "config 2) instead of ONE pass over x (pxsom_assign_sums: 0.36 ms, fixed-point workgroup tables; the "
                         "default since round 3)")
    ap.add_argument("--no-operating-range", action="store_true", help="skip the operating-range legs (data-row / near-tie codebooks)")
    ap.add_argument("--pmc-inner", action="store_true", help=argparse.SUPPRESS)   # the run rocprofv3 wraps
    a = ap.parse_args()
    a.one_pass = not a.two_pass
    return a


def respawn_under_torchrun(args):
    """`--gpus N` from a plain shell: one rank per GPU through torch.distributed.run (the driver's own launch line)."""
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but this box exposes {have} HIP device(s); "
                         f"refusing to run a smaller job under that name")
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    raise SystemExit(subprocess.call(cmd, env=env))


def make_rows(cfg, n_units, rank, dev):
    """This rank's rows in HBM: every rank owns different FOVs / cells."""
    c = cfg["c"]
    tdt = torch.float16 if cfg["dtype"] == "f16" else torch.float32
    if cfg["kind"] == "cell":
        # cells: pixel-cluster counts Poisson(3) / cell_size U(50, 500), 99.9 %-normalised per column
        n = n_units
        g = torch.Generator(device=dev)
        g.manual_seed(2000 + rank)
        x = torch.poisson(torch.full((n, c), 3.0, device=dev), generator=g)
        x.div_(torch.empty((n, 1), device=dev).uniform_(50.0, 500.0, generator=g))
        q = torch.quantile(x[: min(n, 1 << 20)].float(), 0.999, dim=0)
        q[q == 0] = 1.0
        return x.div_(q).to(tdt).contiguous()
    p = cfg["unit_rows"]
    x = torch.empty((n_units * p, c), dtype=tdt, device=dev)
    for f in range(n_units):
        x[f * p:(f + 1) * p] = synth.make_fov_torch(p, c, seed=1000 + rank * n_units + f, device=dev, dtype=tdt)
    return x


def select_launch_counters(rows, counters):
    """From one pass's counter rows of a kernel ({"Dispatch_Id", "Counter_Name", "Counter_Value"} dicts): the counters
    of THE launch over all rows = the dispatches with the largest value of the pass's leading counter, averaged over
    the timed repeats (everything within 10 % of the largest).  The training steps of the bigger configs launch the
    same kernel with the same capped grid: the grid size does not tell them apart, the counters do."""
    by_dispatch = {}
    for r in rows:
        d = by_dispatch.setdefault(r.get("Dispatch_Id"), {})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    lead = counters[-1] if "GRBM_GUI_ACTIVE" in counters else counters[0]
    top = max((d.get(lead, 0.0) for d in by_dispatch.values()), default=0.0)
    chosen = [d for d in by_dispatch.values() if d.get(lead, 0.0) >= 0.9 * top > 0.0]
    out = {}
    for name in counters:
        vals = [d[name] for d in chosen if name in d]
        if vals:
            out[name] = sum(vals) / len(vals)
    return out


def pmc_passes(argv_inner, kernel_substr="bmu_filter"):
    """rocprofv3 counter passes over a short run of this script (its own processes; kernel-trace + pmc only).
    Returns {"FETCH_SIZE": v, ...} = per-dispatch means for the biggest launch of the filter kernel, or {}."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {}
    out = {}
    passes = [["FETCH_SIZE"], ["WRITE_SIZE"], ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_MFMA", "SQ_INSTS_VALU", "GRBM_GUI_ACTIVE"]]
    env = dict(os.environ, TMPDIR="/tmp")
    for counters in passes:
        with tempfile.TemporaryDirectory(dir="/tmp") as td:
            cmd = [exe, "--pmc", *counters, "--kernel-trace", "--output-format", "csv", "-d", td, "-o", "pmc", "--",
                   sys.executable, os.path.abspath(__file__), *argv_inner]
            try:
                subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240,
                               check=True)
            except Exception:
                continue
            rows = []
            for f in glob.glob(os.path.join(td, "**", "*counter_collection.csv"), recursive=True):
                rows += [r for r in csv.DictReader(open(f)) if kernel_substr in r["Kernel_Name"]]
            if not rows:
                continue
            out.update(select_launch_counters(rows, counters))
    return out


def run_cfg1(args):
    """BASELINE.json configs[0] ("cfg1 additionally times the full Python plumbing"): ONE synthetic FOV of
    512 x 512 x 8 channels as the reference stores it (binary64 feather tables: full + 10 % subset + normalisation file) through
    the drop-in functions train_pixel_som -> cluster_pixels -> generate_som_avg_files -- file in, labelled file + weights + CSV out
    -- on the HIP path (`steps` repeats on fresh directories), and ONCE with the oracle standing in for the device entry points
    (the `cpu_baseline` leg: what the reference's own route costs on this box's host cores with pyFlowSOM's arithmetic restated in
    C).  Labels, codebook and mean table of the two legs are compared."""
    import shutil

    import pandas as pd

    from ark_analysis_amd import fov_tables
    from ark_analysis_amd.phenotyping import pixel_som_clustering
    _capi.require_gpu()
    side, c = 512, 8
    n = side * side
    chans = ["chan%d" % i for i in range(c)]
    fovs = ["fov0"]
    x = synth.make_fov_numpy(n, c, seed=501, dtype=np.float64)
    sub = np.sort(np.random.RandomState(7).choice(n, n // 10, replace=False))

    def make_dirs():
        root = tempfile.mkdtemp(prefix="pxsom_cfg1_")
        os.mkdir(os.path.join(root, "pixel_mat_data"))
        os.mkdir(os.path.join(root, "pixel_mat_subsetted"))
        df = pd.DataFrame(x, columns=chans)
        df["fov"] = "fov0"
        df["row_index"] = np.repeat(np.arange(side), side)
        df["column_index"] = np.tile(np.arange(side), side)
        df["label"] = 0
        fov_tables.write_dataframe(df, os.path.join(root, "pixel_mat_data", "fov0.feather"))
        fov_tables.write_dataframe(df.iloc[sub], os.path.join(root, "pixel_mat_subsetted", "fov0.feather"))
        fov_tables.write_dataframe(pd.DataFrame(np.ones((1, c)), columns=chans), os.path.join(root, "post_rowsum_chan_norm.feather"))
        return root

    def run(root, **train_kw):
        stamps = [time.perf_counter()]
        with contextlib.redirect_stdout(io.StringIO()):
            som = pixel_som_clustering.train_pixel_som(fovs, chans, root, num_passes=1, seed=42, **train_kw)
            stamps.append(time.perf_counter())
            pixel_som_clustering.cluster_pixels(fovs, root, som)
            stamps.append(time.perf_counter())
            pixel_som_clustering.generate_som_avg_files(fovs, chans, root, som, data_dir="pixel_mat_data")
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        stamps.append(time.perf_counter())
        labels = fov_tables.read_dataframe(os.path.join(root, "pixel_mat_data", "fov0.feather"))["pixel_som_cluster"].values
        avg = pd.read_csv(os.path.join(root, "pixel_channel_avg_som_cluster.csv"))
        return np.diff(stamps), np.asarray(som.weights.values), labels, avg

    import contextlib
    import io
    legs = {}
    for mode, kw in (("online", {}), ("batch", {"train_mode": "batch"})):     # reference order (exact) / the throughput rule
        times = []
        for i in range(args.warmup + args.steps):
            root = make_dirs()
            try:
                t, w, lab, avg = run(root, **kw)
            finally:
                shutil.rmtree(root, ignore_errors=True)
            if i >= args.warmup:
                times.append(t)
        legs[mode] = (np.mean(np.asarray(times), axis=0), w, lab, avg)
    line = {"metric": "M pixels/sec SOM train+assign through the drop-in pipeline functions, 1 FOV 512x512x8ch, 100-node SOM",
            "unit": "Mpx/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "1 synthetic FOV 512x512x8ch, 10x10 SOM via train_pixel_som -> cluster_pixels -> generate_som_avg_files "
                                   "on binary64 feather tables (BASELINE.json configs[0]: the plumbing config)", "name": "cfg1",
                       "rows": n, "channels": c, "som_nodes": 100, "train_rows": int(len(sub)), "num_passes": 1}}
    t_on, t_ba = legs["online"][0], legs["batch"][0]
    line["value"] = round(n / float(np.sum(t_on)) / 1e6, 3)
    line["ms_per_step"] = round(float(np.sum(t_on)) * 1e3, 2)
    line["phases_ms"] = {"train_mode_online (reference order, exact)": {"train_pixel_som": round(t_on[0] * 1e3, 2), "cluster_pixels": round(t_on[1] * 1e3, 2),
                                                                         "generate_som_avg_files": round(t_on[2] * 1e3, 2)},
                         "train_mode_batch": {"train_pixel_som": round(t_ba[0] * 1e3, 2), "cluster_pixels": round(t_ba[1] * 1e3, 2),
                                              "generate_som_avg_files": round(t_ba[2] * 1e3, 2), "Mpx_per_s": round(n / float(np.sum(t_ba)) / 1e6, 3)}}
    line["roofline"] = None     # host-side plumbing (feather I/O, DataFrames): no kernel of this run is priced against a roofline
    if not args.no_cpu_baseline:
        # the same three calls with the oracle behind ark_analysis_amd.flowsom (test infrastructure, used here as the CPU baseline only)
        from tests import oracle_backend
        from ark_analysis_amd import flowsom
        saved = {}

        def patch(mod, name, value):
            saved.setdefault((mod, name), getattr(mod, name))
            setattr(mod, name, value)
        oracle_backend.install(patch)
        try:
            root = make_dirs()
            try:
                t_cpu, w_cpu, lab_cpu, avg_cpu = run(root)
            finally:
                shutil.rmtree(root, ignore_errors=True)
        finally:
            for (mod, name), value in saved.items():
                setattr(mod, name, value)
        del flowsom
        w_on, lab_on, avg_on = legs["online"][1], legs["online"][2], legs["online"][3]
        line["cpu_baseline"] = {"value": round(n / float(np.sum(t_cpu)) / 1e6, 4), "unit": "Mpx/s", "cores": 1, "kind": "port",
                                "sample": "the whole config (1 FOV: 26 214 training rows x 1 pass in the reference's order, 262 144 rows labelled), "
                                          "same three calls, oracle/pxsom_oracle.c behind ark_analysis_amd.flowsom; once",
                                "phases_ms": {"train_pixel_som": round(t_cpu[0] * 1e3, 2), "cluster_pixels": round(t_cpu[1] * 1e3, 2),
                                              "generate_som_avg_files": round(t_cpu[2] * 1e3, 2)},
                                "gpu_codebook_bit_equal": bool(np.array_equal(w_on, w_cpu)),
                                "gpu_labels_equal": bool(np.array_equal(lab_on, lab_cpu)),
                                "gpu_mean_table_max_rel_err": float(np.max(np.abs(avg_on[chans].values - avg_cpu[chans].values) /
                                                                     np.maximum(np.abs(avg_cpu[chans].values), 1e-300))),
                                "speedup_whole_pipeline": round(float(np.sum(t_cpu)) / float(np.sum(t_on)), 2)}
    print(json.dumps(line))


def main():
    args = parse()
    if args.config == "cfg1":
        if args.gpus != 1:
            raise SystemExit("bench.py: cfg1 is the one-FOV plumbing config (one GPU)")
        return run_cfg1(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "RANK" not in os.environ:
        respawn_under_torchrun(args)
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    _capi.require_gpu()
    # PXSOM_BENCH_DRY_RANKS=1 (tests/test_gpu_bench_multirank.py): every rank on device 0 over a gloo group -- RCCL will not
    # put two ranks on one GPU -- so that the N > 1 code of this script runs on the one-GPU boxes too.  Timings of such
    # a run mean nothing and the line says so.
    dry = os.environ.get("PXSOM_BENCH_DRY_RANKS") == "1"
    if dry:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} has no device (LOCAL_RANK={local_rank}, {torch.cuda.device_count()} visible)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or "RANK" in os.environ     # launched by torch.distributed.run
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if dry:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)

    cfg = CONFIGS[args.config]
    C, XD, YD = cfg["c"], cfg["xdim"], cfg["ydim"]
    K = XD * YD
    esize = 2 if cfg["dtype"] == "f16" else 4
    if cfg["kind"] == "cell":
        n_units = cfg["unit_rows"] // world            # strong scaling: the table is split over the ranks
        units_label = n_units
    else:
        n_units = args.fovs_per_gpu if args.fovs_per_gpu else cfg["units"]
        units_label = n_units
    x_all = make_rows(cfg, n_units, rank, dev)
    n_all = x_all.shape[0]
    stride = int(round(1.0 / cfg["frac"]))
    x_train = x_all if stride == 1 else x_all[::stride].contiguous()   # training subset (every 10th retained pixel)
    n_train = x_train.shape[0]
    g = torch.Generator(device="cpu")
    g.manual_seed(42)
    init_idx = torch.randperm(n_train, generator=g)[:K].to(dev)
    w0 = x_train[init_idx].to(torch.float64).contiguous()
    broadcast_codebook(w0, 0)
    w = w0.clone()
    labels = torch.empty(n_all, dtype=torch.int32, device=dev)
    ws_all = som_device.AssignWorkspace(n_all, C, K, dev) if (not args.one_pass) else som_device.AssignSumsWorkspace(n_all, C, K, dev)
    batch_spec = int(args.batch_steps) if str(args.batch_steps).isdigit() else args.batch_steps
    trainer = BatchSOMTrainer(XD, YD, C, dev, batch_steps=batch_spec)
    sched = trainer.schedule
    k8_sums = torch.empty((K, C), dtype=torch.float64, device=dev)
    k8_counts = torch.empty(K, dtype=torch.int64, device=dev)
    means = torch.empty((K, C), dtype=torch.float64, device=dev)

    def assign_and_mean_table():
        """K7 + K8: BMU label of every row and the per-cluster channel means over every row of every rank -- one
        pass over x (pxsom_assign_sums) with --one-pass, else the BMU search followed by the K8 kernel."""
        if args.one_pass and not use_dist:
            # one process: labels, tables and means from one library call (pxsom_assign_means overwrites its outputs)
            som_device.assign_means(x_all, w, labels, k8_sums, k8_counts, means, ws_all)
            return
        k8_sums.zero_()
        k8_counts.zero_()
        if (not args.one_pass):
            som_device.assign(x_all, w, labels=labels, workspace=ws_all)
            som_device.cluster_sums(x_all, labels, K, sums=k8_sums, counts=k8_counts)
        else:
            som_device.assign_sums(x_all, w, labels=labels, sums=k8_sums, counts=k8_counts, workspace=ws_all)
        if use_dist:
            allreduce_cluster_tables(k8_sums, k8_counts)
        torch.div(k8_sums, k8_counts.clamp(min=1).to(torch.float64).unsqueeze(1), out=means)

    def step():
        # (every step trains from the same first codebook, which stays where it is: the result goes to w -- no copy launch that a
        # run of the pipeline does not have either)
        trainer.train(x_train, w0, num_passes=1, out=w)
        assign_and_mean_table()

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    timer = _capi.KernelTimer(min_rows=n_all)
    ev_train = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                for _ in range(args.steps)]
    ev_k8 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
             for _ in range(args.steps)]
    # the timed region: exactly K steps between two fences, with the HIP events of the roofline's kernel timer (inside the library,
    # on the launch's stream) and nothing else
    with timer:
        t0 = time.perf_counter()
        for i in range(args.steps):
            step()
        fence()
        t1 = time.perf_counter()
        kern_ms, kern_launches = timer.collect()
    # the phases of a step: the same K steps once more, with an event pair around each half (four event records per step cost the
    # stream 10 - 15 us: they are not part of the step and stay out of the timed region)
    for i in range(args.steps):
        ev_train[i][0].record()
        trainer.train(x_train, w0, num_passes=1, out=w)
        ev_train[i][1].record()
        ev_k8[i][0].record()
        assign_and_mean_table()
        ev_k8[i][1].record()
    fence()
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    if use_dist:
        all_reduce_(elapsed, op=dist.ReduceOp.MAX)
    elapsed_s = float(elapsed.item())
    train_ms = float(np.mean([a.elapsed_time(b) for a, b in ev_train]))
    k8_ms = float(np.mean([a.elapsed_time(b) for a, b in ev_k8]))
    # (the one-pass kernel settles its listed rows inside the launch: no list to count afterwards)
    comm_ranks = 0
    exact_rows = som_device.last_exact_rows(ws_all) if (not args.one_pass) or cfg["c"] > 32 or K != 100 else None

    # every rank's own phase times (N > 1: the first multi-GPU line must be readable rank by rank)
    per_rank = None
    if use_dist:
        mine = torch.tensor([train_ms, k8_ms], dtype=torch.float64, device="cpu" if dry else dev)
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        per_rank = {"train_batch": [round(float(g[0]), 4) for g in gathered],
                    "assign_and_mean_table": [round(float(g[1]), 4) for g in gathered]}
        from ark_analysis_amd import distributed as _dm
        from ark_analysis_amd.distributed import native_exchange
        native = native_exchange(None)                                     # (made once per group: cached by now)
        # replicas: every rank must hold the same codebook bit for bit after the timed passes
        wmine = w.detach().cpu() if dry else w.detach().clone()      # (gloo gathers host tensors, RCCL device tensors)
        copies = [torch.zeros_like(wmine) for _ in range(world)]
        dist.all_gather(copies, wmine)
        per_rank["codebooks_equal"] = bool(all(torch.equal(cp, copies[0]) for cp in copies)) and bool(torch.isfinite(wmine).all().item())
        per_rank["kernel_route_agreement"] = getattr(trainer.kernels, "route_agreement", None)
        per_rank["exchange_decision"] = _dm.exchange_report.get(None)
        comm_ranks = world if native is not None else 0
        assert dist.get_world_size() == args.gpus, "process group size and --gpus disagree"
        # the rule's only exchange, timed on its own after the timed region: one [K*C + K] binary64 all-reduce per step, back
        # to back on the route the pass uses -- what the training pass of an N-rank job pays on top of its kernels
        probe = torch.zeros(K * (C + 1), dtype=torch.float64, device=dev)
        reps = 50
        for _ in range(5):
            native.allreduce_sum(probe) if native is not None else all_reduce_(probe)
        fence()
        tp = time.perf_counter()
        for _ in range(reps):
            native.allreduce_sum(probe) if native is not None else all_reduce_(probe)
        fence()
        exch = torch.tensor([(time.perf_counter() - tp) / reps * 1e6], dtype=torch.float64, device="cpu" if dry else dev)
        all_reduce_(exch, op=dist.ReduceOp.MAX)
        per_rank["exchange_us_per_step"] = round(float(exch.item()), 2)
        per_rank["exchange_ms_per_pass"] = round(float(exch.item()) * sched.steps * 1e-3, 4)
        per_rank["exchange_route"] = type(native).__name__ if native is not None else "torch.distributed"
        per_rank["exchange_fused"] = bool(getattr(native, "fused", False))
        # the other in-library route beside it: the one-shot peer-to-peer exchange over HIP IPC blocks (opt-in, PXSOM_EXCHANGE=p2p;
        # validated with two ranks on one device only -- no multi-GPU hardware number exists until a SCALE record holds one)
        p2p_us = (_dm.exchange_report.get(None) or {}).get("p2p_us_per_exchange")    # ("auto": both routes were timed when the communicator was made)
        if not isinstance(native, som_device.P2PComm) and p2p_us is None:
            pc, handles = None, [None] * world
            try:
                pc = som_device.P2PComm(world, rank, K * (C + 1))
            except Exception:   # noqa: BLE001
                pc = None
            dist.all_gather_object(handles, pc.local_handle if pc is not None else None)
            ok = pc is not None and all(h is not None for h in handles)
            if ok:
                try:
                    pc.connect(handles)
                except Exception:   # noqa: BLE001
                    ok = False
            oks = [None] * world
            dist.all_gather_object(oks, bool(ok))
            if all(oks):
                # ONE exchange first, checked on every rank: peer mappings that do not work across devices show up as a
                # bounded wait (4 s) + NaN there, and the timed run of 55 more such waits is not started
                pc.allreduce_sum(probe)
                fence()
                firsts = [None] * world
                dist.all_gather_object(firsts, bool(pc.error_epoch() == 0 and torch.isfinite(probe).all().item()))
                probe.zero_()
                if all(firsts):
                    for _ in range(4):
                        pc.allreduce_sum(probe)
                    fence()
                    tp = time.perf_counter()
                    for _ in range(reps):
                        pc.allreduce_sum(probe)
                    fence()
                    ex2 = torch.tensor([(time.perf_counter() - tp) / reps * 1e6], dtype=torch.float64, device="cpu" if dry else dev)
                    all_reduce_(ex2, op=dist.ReduceOp.MAX)
                    if pc.error_epoch() == 0:
                        p2p_us = round(float(ex2.item()), 2)
                    probe.zero_()
            if pc is not None:
                pc.close()
        per_rank["exchange_us_per_step_p2p"] = p2p_us
    if rank != 0 or args.pmc_inner:
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return

    total_rows = float(n_all) * world * args.steps
    ms_per_step = elapsed_s * 1e3 / args.steps
    value = total_rows / elapsed_s / 1e6
    kern_avg_ms = kern_ms / max(kern_launches, 1)
    bytes_assign = C * esize + 4                       # algorithmic bytes of the assign kernel per row
    bytes_step = C * esize * (1.0 + cfg["frac"]) + 4   # B = C*s*(1 + f*p) + 4, p = 1 pass
    achieved = bytes_assign * n_all / (kern_avg_ms * 1e-3) / 1e9 if kern_launches else 0.0
    step_gbs = bytes_step * n_all / (ms_per_step * 1e-3) / 1e9     # per GPU: every rank moves its own rows
    flops_assign = 2.0 * K * C                         # algorithmic flops per row (x . W^T)
    achieved_tf = flops_assign * n_all / (kern_avg_ms * 1e-3) / 1e12 if kern_launches else 0.0
    mfma_bound = K > 128                               # register-resident filter: HBM; streamed K=400 filter: matrix + VALU pipes
    fast_route = C % 2 == 0 and C <= 32 and 96 < K <= 100   # the register-resident kernels' shapes (csrc/pxsom_assign_filter.hip filter_fast_path)
    out = {
        "metric": "M pixels/sec SOM train+assign, 22-ch 1024^2 FOVs, 100-node SOM",
        "value": round(value, 1), "unit": "Mpx/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": cfg["scaling"], "vs_baseline": None, "dtype": cfg["dtype"], "data": "synthetic",
        "config": {"workload": cfg["desc"].format(u=units_label), "name": args.config,
                   "rows_per_gpu": n_all, "channels": C, "som_nodes": K,
                   "train_mode": "batch", "batch_steps": sched.steps,
                   "batch_schedule": ("two-phase: %d steps over %d phases, widths %s" % (sched.steps, sched.phases, [int(v) for v in np.diff(sched.edges)]))
                   if str(args.batch_steps) == "two-phase" else "equal steps",
                   "train_fraction": cfg["frac"],
                   "num_passes": 1, "step": "train + assign + per-cluster mean table",
                   "parallelism": f"{'row' if cfg['kind'] == 'cell' else 'fov'}-shard x{world}",
                   "rccl_ranks": (0 if dry else world) if use_dist else 0,
                   **({"dry_run": "all ranks on one GPU over gloo: exercises the N > 1 code, timings are meaningless"} if dry else {}),
                   "exchange": ((("in-library peer-to-peer exchange INSIDE the step launches (HIP IPC blocks; last workgroup writes, next step's prologue adds)"
                                  if per_rank.get("exchange_fused") and cfg["kind"] != "cell" else
                                  "in-library peer-to-peer all-reduce (HIP IPC blocks, one launch per rank) behind every step")
                                 if per_rank and per_rank.get("exchange_route") == "P2PComm" else
                                 "in-library RCCL all-reduce behind every step") if comm_ranks else "torch.distributed all-reduce per step")
                   if use_dist else "none (one rank)"},
        "phases_ms": {"train_batch": round(train_ms, 4),
                      "assign_and_mean_table": round(k8_ms, 4),
                      "assign_filter_kernel": round(kern_avg_ms, 4),
                      "assign_exact_rows": exact_rows,
                      "passes_over_x": 2 if (not args.one_pass) else 1,
                      "measured": "event pairs around the halves of the same K steps, run once more after the timed region",
                      **({"per_rank": per_rank} if per_rank else {})},
        "roofline": ({"kernel": "bmu_filter_kernel", "bound": "mfma", "achieved": round(achieved_tf, 1),
                      "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved_tf / MFMA_F16_PEAK_TFLOPS, 4),
                      "traffic": None, "flops_per_row": flops_assign, "hbm_frac": round(achieved / HBM_PEAK_GBS, 4),
                      "rows_per_launch": n_all, "launches_timed": kern_launches,
                      "note": "K = 400 nodes: 400 scores per row put the streamed filter on the matrix + VALU pipes, not on HBM"}
                     if mfma_bound else
                     {"kernel": ("bmu_filter_fast (labels only; the K8 kernel follows)" if fast_route else "bmu_filter_kernel (streamed filter; exact rows and the K8 kernel follow)")
                      if (not args.one_pass) or not fast_route else
                      "bmu_filter_fast<ACC> (labels + per-cluster table in one pass; replaces the filter and the K8 kernel)",
                      "bound": "hbm", "achieved": round(achieved, 1),
                      "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                      "traffic": None, "bytes_per_pixel": bytes_assign,
                      "pixels_per_launch": n_all, "launches_timed": kern_launches}),
        "roofline_step": {"bound": "hbm", "bytes_per_pixel": round(bytes_step, 2), "achieved": round(step_gbs, 1),
                          "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(step_gbs / HBM_PEAK_GBS, 4),
                          "note": "whole step per GPU: SURVEY 8(d) algorithmic bytes / ms_per_step; the training subset (%.0f MB) is the same "
                                  "rows in every timed repeat and fits the 256 MB MALL -- its steps are latency chains, not bandwidth "
                                  "(profiles/r04/assign_sums_size_probe.txt)" % (n_train * C * esize / 1e6)},
    }

    # ---- HBM traffic and matrix-pipe utilisation of the filter kernel: PMC passes of a short run of this script
    pmc = {}
    if world == 1 and not args.no_pmc:
        inner = ["--config", args.config, "--steps", "2", "--warmup", "1", "--batch-steps", str(args.batch_steps),
                 "--no-cpu-baseline", "--no-online", "--no-pmc", "--pmc-inner"] + ([] if args.one_pass else ["--two-pass"])
        if args.fovs_per_gpu:
            inner += ["--fovs-per-gpu", str(args.fovs_per_gpu)]
        pmc = pmc_passes(inner)
    recorded = {}
    rec_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(rec_path):
        try:
            recorded = json.load(open(rec_path))
        except Exception:
            recorded = {}
    if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
        # rocprofv3 reports KiB; FETCH_SIZE counts 128-B requests at 64 B on gfx950 (MI355X_MICROARCH.md "HBM")
        out["roofline"]["traffic"] = round((2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0)
        out["roofline"]["traffic_source"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes inside this run"
    elif args.config == "cfg2" and not mfma_bound and recorded.get(("onepass" if args.one_pass else "bmu_filter") + "_kernel_hbm_bytes_per_launch"):
        out["roofline"]["traffic"] = recorded[("onepass" if args.one_pass else "bmu_filter") + "_kernel_hbm_bytes_per_launch"]
        out["roofline"]["traffic_source"] = "profiles/pmc_traffic.json (recorded; no in-run PMC pass)"
    if "SQ_VALU_MFMA_BUSY_CYCLES" in pmc and pmc.get("GRBM_GUI_ACTIVE"):
        kernel_cycles = pmc["GRBM_GUI_ACTIVE"] / N_XCDS     # cycles the kernel was resident (per XCD)
        out["mfma_util"] = {"kernel": out["roofline"]["kernel"].split(" ")[0],
                            "value": round(pmc["SQ_VALU_MFMA_BUSY_CYCLES"] / (N_SIMDS * kernel_cycles), 4),
                            "mfma_busy_cycles": pmc["SQ_VALU_MFMA_BUSY_CYCLES"], "kernel_cycles": round(kernel_cycles),
                            "mfma_insts": pmc.get("SQ_INSTS_MFMA"), "valu_insts": pmc.get("SQ_INSTS_VALU"),
                            "source": "rocprofv3 --pmc inside this run: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x "
                                      "GRBM_GUI_ACTIVE / 8 XCDs)"}
    elif recorded.get(("onepass" if args.one_pass else "bmu_filter") + "_kernel_mfma_util") and args.config == "cfg2":
        out["mfma_util"] = {"kernel": out["roofline"]["kernel"].split(" ")[0],
                            "value": recorded[("onepass" if args.one_pass else "bmu_filter") + "_kernel_mfma_util"],
                            "source": "profiles/pmc_traffic.json (recorded; no in-run PMC pass)"}

    if world == 1 and args.config == "cfg2" and not args.no_operating_range:
        # Operating range of "bit-exact at HBM speed" (tests/tools/robustness_sweep.py holds the full sweep): the synthetic
        # workload lists 0.02 % of its rows for the exact path; codebooks made of data rows (distance-0 matches) and
        # codebooks with node pairs 1e-2 apart list far more.  Same rows, labels only (pxsom_assign), HIP-event timed.
        def timed_assign(wcb, reps=3):
            ws_r = som_device.AssignWorkspace(n_all, C, K, dev)
            lab_r = torch.empty(n_all, dtype=torch.int32, device=dev)
            som_device.assign(x_all, wcb, labels=lab_r, workspace=ws_r)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                som_device.assign(x_all, wcb, labels=lab_r, workspace=ws_r)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            listed = som_device.last_exact_rows(ws_r)
            # the default path of the step on the same codebook: labels + tables + means in one pass (its listed rows are
            # settled inside the launch, one wave per row)
            ws_m = som_device.AssignSumsWorkspace(n_all, C, K, dev)
            som_device.assign_means(x_all, wcb, lab_r, k8_sums, k8_counts, means, ws_m)
            e0.record()
            for _ in range(reps):
                som_device.assign_means(x_all, wcb, lab_r, k8_sums, k8_counts, means, ws_m)
            e1.record()
            torch.cuda.synchronize()
            return {"assign_ms": round(ms, 4), "listed_rows": listed, "listed_frac": round(listed / n_all, 6),
                    "Gpx_per_s": round(n_all / ms / 1e6, 2), "labels_and_mean_table_one_pass_ms": round(e0.elapsed_time(e1) / reps, 4)}
        near = w.clone()
        gen = torch.Generator(device=dev)
        gen.manual_seed(5)
        near[K // 2:] = near[:K - K // 2] * (1.0 + 1e-2 * torch.randn((K - K // 2, C), dtype=torch.float64, device=dev, generator=gen))
        out["operating_range"] = {"trained codebook": timed_assign(w), "codebook = data rows": timed_assign(w0),
                                  "node pairs 1e-2 apart": timed_assign(near),
                                  "note": "labels of all rows, bit-equal to the oracle in every case (tests); time follows the rows the "
                                          "filter lists for the exact binary64 path"}
        # Neighbouring rows that share their label -- what images look like and the synthetic FOVs do not: the same rows in runs of
        # 64 equal labels (sorted by label, the runs shuffled).  The 16 rows of one LDS atomic then hit the same table words; the
        # one-pass kernel sums such tiles along the row axis first (scripts/debug/label_coherence_probe.py has the other run lengths).
        order = torch.argsort(labels.long(), stable=True)
        runs = n_all // 64
        idx = (torch.randperm(runs, device=dev).unsqueeze(1) * 64 + torch.arange(64, device=dev).unsqueeze(0)).reshape(-1)
        x_keep = x_all
        x_all = x_all[order[idx]].contiguous()
        del order, idx
        out["operating_range"]["rows in runs of 64 equal labels"] = timed_assign(w)
        x_all = x_keep
        del x_keep

    if args.config == "cfg5":
        # "+ consensus meta-cluster" (BASELINE.json configs[4]): Ward on the K x C mean table on the host (the
        # reference's PixieConsensusCluster step), then the K-entry lookup over the labels in HBM.  Outside the timed
        # step: reported beside it.
        import pandas as pd
        from ark_analysis_amd.phenotyping.cluster_helpers import PixieConsensusCluster
        chans = ["chan%d" % j for j in range(C)]
        avg = pd.DataFrame(means.cpu().numpy(), columns=chans)
        avg.insert(0, "pixel_som_cluster", np.arange(1, K + 1))
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "pixel_channel_avg_som_cluster.csv")
            avg.to_csv(path, index=False)
            tt = time.perf_counter()
            cc = PixieConsensusCluster("pixel", path, chans, max_k=20, cap=3)
            cc.scale_data()
            np.random.seed(42)
            cc.run_consensus_clustering()
            cc.generate_som_to_meta_map()
            t_ward = time.perf_counter() - tt
        lut = torch.from_numpy(cc.lookup_table().astype(np.int32)).to(dev)
        meta = torch.empty_like(labels)
        som_device.relabel(labels, lut, out=meta)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        som_device.relabel(labels, lut, out=meta)
        e1.record()
        torch.cuda.synchronize()
        out["consensus"] = {"max_k": 20, "ward_host_ms": round(t_ward * 1e3, 2), "relabel_kernel_ms": round(e0.elapsed_time(e1), 4),
                            "meta_clusters_found": int(torch.unique(meta).numel()),
                            "note": "K x C table on the host (as the reference), K-entry lookup in HBM; not part of the timed step"}

    if world == 1 and args.config != "cfg2" and not args.no_cpu_baseline:
        # The other configurations: the same oracle (online FlowSOM training + reference-shaped BMU search + per-cluster
        # sums, one host core) on a BOUNDED sample -- about 10 s of training steps and 10 s of row labelling -- scaled
        # linearly to the configuration's rows; the GPU labels of the sampled rows are compared on the way.
        from tests import oracle_binding as ob
        rr = default_radius_range(XD, YD)
        s_train = int(min(n_train, max(20_000, 8e9 / (K * C))))
        s_assign = int(min(n_all, max(50_000, 1.2e10 / (K * C))))
        xt = x_train[:s_train].cpu().numpy().astype(np.float64)
        w0h = w0.cpu().numpy()
        order = np.random.RandomState(7).randint(0, s_train, size=s_train).astype(np.int64)
        tt = time.perf_counter()
        oracle_w = ob.som_online(xt, w0h, XD, YD, 1, (0.05, 0.01), rr, order)
        t_train = time.perf_counter() - tt
        xs = x_all[:s_assign].cpu().numpy().astype(np.float64)
        tt = time.perf_counter()
        lab_cpu, _ = ob.map_data_to_nodes(oracle_w, xs, column_major_copy=True)
        t_assign = time.perf_counter() - tt
        tt = time.perf_counter()
        ob.cluster_sums(xs, lab_cpu, K)
        t_means = time.perf_counter() - tt
        cpu_s = t_train * (n_train / s_train) + (t_assign + t_means) * (n_all / s_assign)
        lab_gpu, _ = som_device.assign(x_all[:s_assign], torch.from_numpy(oracle_w).to(dev))
        out["cpu_baseline"] = {
            "value": round(n_all / cpu_s / 1e6, 4), "unit": "Mpx/s" if cfg["kind"] == "pixel" else "M cells/s", "cores": 1,
            "host_cores": os.cpu_count(), "kind": "port",
            "sample": f"oracle online FlowSOM training on {s_train} of {n_train} training rows ({t_train:.2f} s) + "
                      f"reference-shaped BMU search ({t_assign:.2f} s) and per-cluster sums ({t_means:.2f} s) on {s_assign} of "
                      f"{n_all} rows, each scaled linearly; fp64, 1 thread",
            "gpu_labels_equal_on_sample": bool(np.array_equal(lab_gpu.cpu().numpy(), lab_cpu))}

    if world == 1 and args.config == "cfg2":
        oracle_w = None
        order = None
        rr = default_radius_range(XD, YD)
        if not args.no_cpu_baseline:
            from tests import oracle_binding as ob
            xt = x_train.cpu().numpy().astype(np.float64)
            w0h = w0.cpu().numpy()
            # the timed training mode at FULL size against its oracle (the bench's own codebook, last timed step)
            tt = time.perf_counter()
            want_b = ob.som_batch_sched(xt, w0h, XD, YD, 1, (0.05, 0.01), rr, sched.phases, sched.edges)
            t_b = time.perf_counter() - tt
            got_b = w.cpu().numpy()
            rel = float(np.max(np.abs(got_b - want_b) / np.maximum(np.abs(want_b), 1e-300)))
            out["batch_train"] = {"rows": n_train, "steps": sched.steps,
                                  "codebook_matches_oracle_rtol_1e-9": bool(np.allclose(got_b, want_b, rtol=1e-9, atol=0)),
                                  "codebook_bit_equal_to_oracle": bool(np.array_equal(got_b, want_b)),
                                  "max_rel_err": rel, "oracle_s": round(t_b, 2)}
            rs = np.random.RandomState(7)
            order = rs.randint(0, n_train, size=n_train).astype(np.int64)
            tt = time.perf_counter()
            oracle_w = ob.som_online(xt, w0h, XD, YD, 1, (0.05, 0.01), rr, order)
            t_train = time.perf_counter() - tt
            # quality of the timed mode: mean quantisation error (distance to the BMU) over ALL rows of the batch
            # codebook against the codebook the online (reference-order) oracle reaches from the same initial nodes
            def mean_qe(codebook):
                _, d = som_device.assign(x_all, torch.from_numpy(np.ascontiguousarray(codebook)).to(dev), want_dists=True)
                return float(d.mean().item())
            qe_batch, qe_online = mean_qe(got_b), mean_qe(oracle_w)
            out["batch_train"]["quantisation_error"] = {"batch": qe_batch, "online_oracle": qe_online,
                                                         "batch_vs_online_pct": round((qe_batch / qe_online - 1.0) * 100.0, 3),
                                                         "rows": n_all}
            # the reference labels one FOV table per call (cluster_pixels): so does this leg, FOV by FOV, over
            # 8 of the FOVs (~10 s of host work together with the training leg)
            P = cfg["unit_rows"]
            f_s = min(n_units, 8)
            n_s = f_s * P
            t_assign = t_means = 0.0
            lab_parts = []
            for f in range(f_s):
                xs = x_all[f * P:(f + 1) * P].cpu().numpy().astype(np.float64)
                tt = time.perf_counter()
                lab_f, _ = ob.map_data_to_nodes(oracle_w, xs, column_major_copy=True)
                t_assign += time.perf_counter() - tt
                tt = time.perf_counter()
                ob.cluster_sums(xs, lab_f, K)
                t_means += time.perf_counter() - tt
                lab_parts.append(lab_f)
            lab_cpu = np.concatenate(lab_parts)
            cpu_s = t_train + (t_assign + t_means) * (n_all / n_s)
            # free full-size check: GPU labels for the same codebook on the same sample
            wd = torch.from_numpy(oracle_w).to(dev)
            lab_gpu, _ = som_device.assign(x_all[:n_s], wd)
            labels_equal = bool(np.array_equal(lab_gpu.cpu().numpy(), lab_cpu))
            # The reference's own process-parallel mode beside it (cluster_pixels(multiprocess=True, batch_size=5),
            # pixel_som_clustering.py:140, :257-271): min(5, cores) worker processes, one FOV table each, the oracle's BMU
            # search + per-cluster sums; training stays the sequential loop it is.  Workers are fresh interpreters (numpy +
            # the oracle library only); the span is first compute start -> last compute end.
            mp_field = None
            try:
                procs_n = min(5, os.cpu_count() or 1, f_s)
                with tempfile.TemporaryDirectory(prefix="pxsom_cpu_mp_") as td:
                    np.save(os.path.join(td, "w.npy"), oracle_w)
                    for f in range(procs_n):
                        np.save(os.path.join(td, "x%d.npy" % f), x_all[f * P:(f + 1) * P].cpu().numpy())
                    code = ("import sys, time, numpy as np; sys.path.insert(0, %r); from tests import oracle_binding as ob; "
                            "w = np.load(sys.argv[1]); x = np.load(sys.argv[2]).astype(np.float64); t0 = time.time(); "
                            "lab, _ = ob.map_data_to_nodes(w, x, column_major_copy=True); ob.cluster_sums(x, lab, w.shape[0]); "
                            "print(t0, time.time())" % os.path.dirname(os.path.abspath(__file__)))
                    ps = [subprocess.Popen([sys.executable, "-c", code, os.path.join(td, "w.npy"), os.path.join(td, "x%d.npy" % f)],
                                           stdout=subprocess.PIPE, text=True) for f in range(procs_n)]
                    spans = [tuple(float(v) for v in p_.communicate(timeout=600)[0].split()) for p_ in ps]
                span = max(e for _, e in spans) - min(b for b, _ in spans)
                label_rate = procs_n * P / span                                   # pixels / s with procs_n workers
                mp_s = t_train + n_all / label_rate
                mp_field = {"processes": procs_n, "value": round(n_all / mp_s / 1e6, 4), "unit": "Mpx/s",
                            "labelling_Mpx_per_s": round(label_rate / 1e6, 3),
                            "sample": f"{procs_n} worker processes x 1 FOV table each ({span:.2f} s for {procs_n * P} pixels), "
                                      f"training as above ({t_train:.2f} s, sequential by construction)"}
            except Exception as err:   # noqa: BLE001 -- a baseline, not the product: say so and go on
                mp_field = {"error": str(err)[:200]}
            out["cpu_baseline"] = {
                "multiprocess": mp_field,
                "value": round(n_all / cpu_s / 1e6, 4), "unit": "Mpx/s", "cores": 1,
                "host_cores": os.cpu_count(), "kind": "port",
                "sample": f"oracle online FlowSOM training on the full {n_train}-row training subset "
                          f"({t_train:.2f} s) + reference-shaped BMU search ({t_assign:.2f} s) and per-cluster "
                          f"sums ({t_means:.2f} s) on {n_s} of {n_all} pixels (one FOV per call, as cluster_pixels does), scaled linearly; fp64, 1 thread",
                "gpu_labels_equal_on_sample": labels_equal}
        if not args.no_online:
            if order is None:
                order = np.random.RandomState(7).randint(0, n_train, size=n_train).astype(np.int64)
            od = torch.from_numpy(order).to(dev)
            wo = w0.clone()
            som_device.train_online(x_train, wo, XD, YD, 1, (0.05, 0.01), rr, od)  # warm
            wo.copy_(w0)
            torch.cuda.synchronize()
            tt = time.perf_counter()
            som_device.train_online(x_train, wo, XD, YD, 1, (0.05, 0.01), rr, od)
            torch.cuda.synchronize()
            t_on = time.perf_counter() - tt
            out["online_train"] = {"ms": round(t_on * 1e3, 2), "steps": n_train,
                                   "steps_per_s": round(n_train / t_on, 1),
                                   "note": "exact reference-order mode; latency-bound, replicas only"}
            if oracle_w is not None:
                out["online_train"]["codebook_bit_equal_to_oracle"] = bool(
                    np.array_equal(wo.cpu().numpy(), oracle_w))
    print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
