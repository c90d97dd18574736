#!/usr/bin/env python
"""bench.py -- Pixie pixel-SOM train + assign throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one pass of the hot path over one batch of synthetic input, resident in HBM before
the clock starts ("normalised pixel matrix resident" -> "labels + codebook + mean table
resident"):  batch-mode SOM training (1 pass over the 10 % training subset, `--batch-steps` mini-batch
steps, statistics all-reduced over RCCL when N > 1)  +  BMU assignment of every pixel  +  the per-cluster
mean-expression table over all pixels (sums/counts all-reduced once when N > 1).
Workload at N = 1: BASELINE.json configs[1] (10 FOVs 1024x1024x22 fp32, 10x10 SOM); weak scaling:
every rank holds its own `--fovs-per-gpu` FOVs.  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline     dominant kernel = bmu_filter_kernel over all pixels; achieved = 92 B/pixel
               (22*4 read + 4 label written, DESIGN.md "K7") * pixels / its HIP-event duration.
  cpu_baseline the oracle (port of the reference algorithm: online FlowSOM training on the same
               training subset + reference-shaped BMU search) on ONE host core; bounded sample.
  online_train the exact-online (reference-order) training kernel on the same subset, checked
               against the oracle's codebook from the cpu_baseline leg (full-size parity).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from ark_analysis_amd import _capi, som_device, synth  # noqa: E402
from ark_analysis_amd.distributed import (BatchSOMTrainer, allreduce_cluster_tables,  # noqa: E402
                                          broadcast_codebook)
from ark_analysis_amd.flowsom import default_radius_range  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PIXELS_PER_FOV = 1024 * 1024
CHANNELS = 22
XDIM = YDIM = 10
BYTES_PER_PIXEL_ASSIGN = CHANNELS * 4 + 4   # algorithmic bytes of the assign kernel (fp32 in, i32 out)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--fovs-per-gpu", type=int, default=10)
    ap.add_argument("--batch-steps", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-online", action="store_true")
    return ap.parse_args()


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    _capi.require_gpu()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or "RANK" in os.environ     # launched by torch.distributed.run
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)

    F, P, C, K = args.fovs_per_gpu, PIXELS_PER_FOV, CHANNELS, XDIM * YDIM
    n_all = F * P
    # ---- synthetic input, generated in HBM; every rank owns different FOVs
    x_all = torch.empty((n_all, C), dtype=torch.float32, device=dev)
    for f in range(F):
        x_all[f * P:(f + 1) * P] = synth.make_fov_torch(P, C, seed=1000 + rank * F + f, device=dev)
    x_train = x_all[::10].contiguous()            # 10 % training subset (every 10th retained pixel)
    n_train = x_train.shape[0]
    g = torch.Generator(device="cpu")
    g.manual_seed(42)
    init_idx = torch.randperm(n_train, generator=g)[:K].to(dev)
    w0 = x_train[init_idx].to(torch.float64).contiguous()
    broadcast_codebook(w0, 0)
    w = w0.clone()
    labels = torch.empty(n_all, dtype=torch.int32, device=dev)
    ws_all = som_device.AssignWorkspace(n_all, C, K, dev)
    trainer = BatchSOMTrainer(XDIM, YDIM, C, dev, batch_steps=args.batch_steps)
    k8_sums = torch.empty((K, C), dtype=torch.float64, device=dev)
    k8_counts = torch.empty(K, dtype=torch.int64, device=dev)
    means = torch.empty((K, C), dtype=torch.float64, device=dev)

    def mean_table():
        """K8: per-cluster channel means over every pixel of every rank."""
        k8_sums.zero_()
        k8_counts.zero_()
        som_device.cluster_sums(x_all, labels, K, sums=k8_sums, counts=k8_counts)
        if use_dist:
            allreduce_cluster_tables(k8_sums, k8_counts)
        torch.div(k8_sums, k8_counts.clamp(min=1).to(torch.float64).unsqueeze(1), out=means)

    def step():
        w.copy_(w0)
        trainer.train(x_train, w, num_passes=1)
        som_device.assign(x_all, w, labels=labels, workspace=ws_all)
        mean_table()

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
    with timer:
        t0 = time.perf_counter()
        for i in range(args.steps):
            w.copy_(w0)
            ev_train[i][0].record()
            trainer.train(x_train, w, num_passes=1)
            ev_train[i][1].record()
            som_device.assign(x_all, w, labels=labels, workspace=ws_all)
            ev_k8[i][0].record()
            mean_table()
            ev_k8[i][1].record()
        fence()
        t1 = time.perf_counter()
        kern_ms, kern_launches = timer.collect()
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed_s = float(elapsed.item())
    train_ms = float(np.mean([a.elapsed_time(b) for a, b in ev_train]))
    k8_ms = float(np.mean([a.elapsed_time(b) for a, b in ev_k8]))
    exact_rows = som_device.last_exact_rows(ws_all)

    if rank != 0:
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return

    total_pixels = float(n_all) * world * args.steps
    ms_per_step = elapsed_s * 1e3 / args.steps
    value = total_pixels / elapsed_s / 1e6
    kern_avg_ms = kern_ms / max(kern_launches, 1)
    achieved = BYTES_PER_PIXEL_ASSIGN * n_all / (kern_avg_ms * 1e-3) / 1e9 if kern_launches else 0.0
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_path):
        try:
            traffic = json.load(open(pmc_path)).get("bmu_filter_kernel_hbm_bytes_per_launch")
        except Exception:
            traffic = None
    out = {
        "metric": "M pixels/sec SOM train+assign, 22-ch 1024^2 FOVs, 100-node SOM",
        "value": round(value, 1), "unit": "Mpx/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{F} FOVs 1024x1024x22ch fp32 per GPU, 10x10 SOM "
                               f"(BASELINE.json configs[1] at N=1)",
                   "fovs_per_gpu": F, "pixels_per_gpu": n_all, "channels": C, "som_nodes": K,
                   "train_mode": "batch", "batch_steps": args.batch_steps, "train_fraction": 0.1,
                   "num_passes": 1, "step": "train + assign + per-cluster mean table",
                   "parallelism": f"fov-shard x{world}"},
        "phases_ms": {"train_batch": round(train_ms, 4),
                      "assign_filter_kernel": round(kern_avg_ms, 4),
                      "assign_exact_rows": exact_rows,
                      "mean_table": round(k8_ms, 4)},
        "roofline": {"kernel": "bmu_filter_kernel", "bound": "hbm", "achieved": round(achieved, 1),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "traffic": traffic, "bytes_per_pixel": BYTES_PER_PIXEL_ASSIGN,
                     "pixels_per_launch": n_all, "launches_timed": kern_launches},
    }

    if world == 1:
        oracle_w = None
        order = None
        if not args.no_cpu_baseline:
            from tests import oracle_binding as ob
            xt = x_train.cpu().numpy().astype(np.float64)
            w0h = w0.cpu().numpy()
            rs = np.random.RandomState(7)
            order = rs.randint(0, n_train, size=n_train).astype(np.int64)
            rr = default_radius_range(XDIM, YDIM)
            tt = time.perf_counter()
            oracle_w = ob.som_online(xt, w0h, XDIM, YDIM, 1, (0.05, 0.01), rr, order)
            t_train = time.perf_counter() - tt
            # the reference labels one FOV table per call (cluster_pixels): so does this leg, FOV by FOV, over
            # 8 of the FOVs (~10 s of host work together with the training leg)
            f_s = min(F, 8)
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
            out["cpu_baseline"] = {
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
            rr = default_radius_range(XDIM, YDIM)
            som_device.train_online(x_train, wo, XDIM, YDIM, 1, (0.05, 0.01), rr, od)  # warm
            wo.copy_(w0)
            torch.cuda.synchronize()
            tt = time.perf_counter()
            som_device.train_online(x_train, wo, XDIM, YDIM, 1, (0.05, 0.01), rr, od)
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
