R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r3_tune
cd $R
timeout 600 python -m pytest tests/test_gpu_preprocessing.py tests/test_gpu_schedule.py -x -q > gpurun_out/r3_tune/pytest.log 2>&1; tail -3 gpurun_out/r3_tune/pytest.log
for env in "PXSOM_STEP_WGS_PER_CU=2" "PXSOM_STEP_WGS_PER_CU=1" "PXSOM_STEP_WGS_PER_CU=2 PXSOM_STEP_TPW=2" "PXSOM_STEP_WGS_PER_CU=2 PXSOM_STEP_TPW=1"; do
env $env python bench.py --steps 5 --warmup 2 --no-pmc --no-cpu-baseline --no-online --one-pass 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$env',d['value'],d['ms_per_step'],d['phases_ms'])" | tee -a gpurun_out/r3_tune/bench.txt
done
python scripts/preprocess_kernels.py --reps 5 | tee gpurun_out/r3_tune/pre.json
BENCH_ARGS="--one-pass" bash scripts/jobs/r3_trace.sh
cp gpurun_out/r3_trace/kernel_trace_stats.txt gpurun_out/r3_tune/
