"""Where does a whole batch run on the GPU part ways with the oracle's?  Step by step, both trajectories."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from ark_analysis_amd import som_device as sd, synth
from ark_analysis_amd.distributed import batch_schedule
from ark_analysis_amd.flowsom import default_radius_range
from ark_analysis_amd.schedule import BatchSchedule
from tests import oracle_binding as ob

c, n, k = 8, 2500, 100
sch = BatchSchedule.two_phase(head_steps=3, tail_steps=5, head_ratio=0.5, tail_phases_per_step=2)
print("phases", sch.phases, "edges", sch.edges)
x = synth.make_fov_numpy(max(n, 2 * k), c, seed=51, dtype=np.float32)[:n]
rs = np.random.RandomState(7)
w0s = synth.make_fov_numpy(4 * k, c, seed=52, dtype=np.float64)
w0 = np.ascontiguousarray(w0s[rs.choice(w0s.shape[0], size=k, replace=False)].astype(np.float64))
w0[k - 2] = w0[3]
gpu = torch.device("cuda:0")
xd = torch.from_numpy(x).to(gpu)
x64 = x.astype(np.float64)
rr = default_radius_range(10, 10)
total = sch.steps
st = sd.BatchTrainState(n, c, 10, 10, sch, gpu, dtype=xd.dtype)
st.wbuf[0].copy_(torch.from_numpy(w0))
w_or = w0.copy()
s_or = cnt_or = None
for g in range(total):
    sd.batch_train_steps(xd, st, g, g + 1, total, (0.05, 0.01), rr)
    w_g = st.wbuf[g % 2].cpu().numpy()
    if g > 0:
        thr, alpha = batch_schedule(sch.position(g - 1), sch.phases, (0.05, 0.01), rr)
        w_or = ob.batch_update(w_or, 10, 10, s_or, cnt_or, thr, alpha)
    rows = x64[sch.rows_of_step(n, g)]
    lab_or, _ = ob.map_data_to_nodes(w_or, rows)
    s_or, cnt_or = ob.cluster_sums(rows, lab_or, k)
    lab_g, _ = ob.map_data_to_nodes(w_g, rows)
    ring = st.ring[g % 3].cpu().numpy()
    d = np.abs(w_g - w_or)
    rel = d / np.maximum(np.abs(w_or), 1e-300)
    print(f"step {g}: rows {len(rows)} thr {batch_schedule(sch.position(g - 1), sch.phases, (0.05, 0.01), rr) if g else None} "
          f"max rel diff W {rel.max():.3e} at node {np.unravel_index(rel.argmax(), rel.shape)}; labels differ (gpu W vs oracle W): {(lab_g != lab_or).sum()}; "
          f"gpu counts == oracle-on-gpu-W counts: {np.array_equal(ring[k*c:], np.bincount(lab_g - 1, minlength=k).astype(float))}")
    if g > 0 and rel.max() > 0:
        bad = np.argwhere(rel > 1e-12)
        print("   nodes off:", sorted(set(bad[:, 0].tolist()))[:20], " dup nodes 3/98 equal on gpu:", np.array_equal(w_g[3], w_g[98]), "oracle:", np.array_equal(w_or[3], w_or[98]))
