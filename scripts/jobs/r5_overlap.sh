# Round 5: overlap across the step boundary (tail steps on two streams, flag instead of stream order): first a bounded smoke run, then parity,
# then the same-box A/B (PXSOM_STEP_OVERLAP=0 / 1)
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r5_overlap; mkdir -p $O
timeout 300 python - > $O/smoke.txt 2>&1 <<'PY'
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from ark_analysis_amd import som_device as sd, synth
from ark_analysis_amd.schedule import BatchSchedule
from ark_analysis_amd.flowsom import default_radius_range
from tests import oracle_binding as ob
dev = torch.device("cuda:0")
n, c, k = 200_003, 22, 100
x = synth.make_fov_numpy(n, c, seed=5, dtype=np.float32)
x = (np.round(x.astype(np.float64) * 4096.0) / 4096.0).astype(np.float32)
rs = np.random.RandomState(1)
w0 = x[rs.choice(n, k, replace=False)].astype(np.float64)
sch = BatchSchedule.two_phase()
xd = torch.from_numpy(x).to(dev)
rr = default_radius_range(10, 10)
st = sd.BatchTrainState(n, c, 10, 10, sch, dev, dtype=xd.dtype)
for rep in range(3):
    st.wbuf[0].copy_(torch.from_numpy(w0))
    t0 = time.perf_counter()
    sd.batch_train_steps(xd, st, 0, sch.steps, sch.steps, (0.05, 0.01), rr)
    w = torch.empty((k, c), dtype=torch.float64, device=dev)
    sd.batch_train_finish(st, sch.steps, sch.steps, (0.05, 0.01), rr, w)
    torch.cuda.synchronize()
    print("pass", rep, "%.3f ms" % ((time.perf_counter() - t0) * 1e3), "finite", bool(torch.isfinite(w).all()))
want = ob.som_batch_sched(x.astype(np.float64), w0, 10, 10, 1, (0.05, 0.01), rr, sch.phases, sch.edges)
print("bit-equal to the oracle:", np.array_equal(w.cpu().numpy(), want))
PY
cat $O/smoke.txt
grep -q "bit-equal to the oracle: True" $O/smoke.txt || { echo "smoke failed: stopping"; exit 0; }
timeout 2400 python -m pytest tests/test_gpu_schedule.py tests/test_gpu_fuzz_parity.py tests/test_pipeline_dropin.py tests/test_gpu_som_kernels.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for v in 1 0 1 0; do PXSOM_STEP_OVERLAP=$v python bench.py --no-cpu-baseline --no-online --no-pmc --no-operating-range 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('PXSOM_STEP_OVERLAP=$v', d['value'], d['ms_per_step'], d['phases_ms']['train_batch'], d['phases_ms']['assign_filter_kernel'], d.get('batch_train'))"; done | tee $O/summary.txt
