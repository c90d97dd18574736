#!/usr/bin/env python
"""Operating range of the "bit-exact at HBM speed" assignment: the BMU filter sends
every row whose two best scores are closer than its error bound to the exact binary64 path, so its speed depends on
how many rows are such near-ties.  The synthetic bench workload lists 0.02 % of its rows; real MIBI tables hold many
exact zeros, duplicate rows and -- early in training -- near-identical nodes.  This sweep makes the data
progressively nastier and reports listed-row fraction and assign time (config 2 shape: 10 x 1024^2 x 22 fp32, K = 100):
  quantise   values rounded to multiples of q (coarser q -> more exact ties between rows and nodes)
  codebook   trained (batch rule) / rows of the data themselves (every such row is at distance 0 of a node) /
             with groups of near-identical nodes (node + 1e-7 relative noise)
Labels are checked against the oracle on a 50 k-row sample in every setting."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ark_analysis_amd import som_device as sd, synth          # noqa: E402
from ark_analysis_amd.distributed import BatchSOMTrainer      # noqa: E402
from tests import oracle_binding as ob                        # noqa: E402

dev = torch.device("cuda:0")
F, P, C, K = 10, 1024 * 1024, 22, 100
base = torch.cat([synth.make_fov_torch(P, C, seed=1000 + f, device=dev) for f in range(F)])
n = base.shape[0]


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


rows = []
for q in (0.0, 2.0 ** -12, 2.0 ** -8, 2.0 ** -6, 2.0 ** -4):
    x = base if q == 0.0 else (torch.round(base / q) * q).contiguous()
    sub = x[::10].contiguous()
    g = torch.Generator(device="cpu")
    g.manual_seed(42)
    w0 = sub[torch.randperm(sub.shape[0], generator=g)[:K].to(dev)].double().contiguous()
    trained = w0.clone()
    BatchSOMTrainer(10, 10, C, dev, batch_steps=64).train(sub, trained, 1)
    variants = [("trained", trained), ("data rows", w0)]
    if q == 0.0:
        # pairs of nodes a relative distance eps apart: the smaller eps, the more rows cannot be told apart by the filter
        for eps in (1e-2, 1e-3, 3e-4, 1e-4, 1e-5, 1e-7):
            near = trained.clone()
            near[50:] = near[:50] * (1.0 + eps * torch.randn((50, C), dtype=torch.float64, device=dev))
            variants.append(("node pairs %.0e apart" % eps, near))
    for name, w in variants:
        labels = torch.empty(n, dtype=torch.int32, device=dev)
        ws = sd.AssignWorkspace(n, C, K, dev)
        ms = timed(lambda: sd.assign(x, w, labels=labels, workspace=ws))
        listed = sd.last_exact_rows(ws)
        idx = torch.randperm(n, device=dev)[:50_000]
        want, _ = ob.map_data_to_nodes(w.cpu().numpy(), x[idx].double().cpu().numpy())
        ok = bool(np.array_equal(labels[idx].cpu().numpy(), want))
        rows.append(dict(quantum=q, codebook=name, listed_rows=listed, listed_frac=listed / n, assign_ms=round(ms, 4),
                         gpx_per_s=round(n / ms / 1e6, 2), labels_equal_oracle_on_sample=ok))
        print(json.dumps(rows[-1]), flush=True)
assert all(r["labels_equal_oracle_on_sample"] for r in rows)
