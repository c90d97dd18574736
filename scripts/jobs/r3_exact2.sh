R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r3_exact2
timeout 900 python -m pytest tests/test_gpu_som_kernels.py tests/test_gpu_schedule.py tests/test_gpu_fuzz_parity.py -x -q -k "batch or schedule or reproducib or fuzz_batch" > gpurun_out/r3_exact2/pytest.log 2>&1; tail -3 gpurun_out/r3_exact2/pytest.log
for i in 1 2; do python bench.py --steps 10 --warmup 2 --no-pmc --no-cpu-baseline --no-online --no-operating-range 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['phases_ms'])" | tee -a gpurun_out/r3_exact2/bench.txt; done
bash scripts/jobs/r3_trace.sh
cp gpurun_out/r3_trace/kernel_trace_stats.txt gpurun_out/r3_exact2/
PXSOM_FUZZ_CASES=400 PXSOM_FUZZ_SEED=77 timeout 1200 python -m pytest tests/test_gpu_fuzz_parity.py -q -x -k batch 2>&1 | tail -2
