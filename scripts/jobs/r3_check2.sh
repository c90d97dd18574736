# Round 3: after the branch-free accumulate / step-size dependent tiles / BMU-only update path
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r3_check2
cd $R
timeout 1200 python -m pytest tests/test_gpu_som_kernels.py tests/test_gpu_schedule.py -x -q > gpurun_out/r3_check2/pytest.log 2>&1; tail -4 gpurun_out/r3_check2/pytest.log
timeout 600 python -m pytest tests/test_gpu_fuzz_parity.py -x -q > gpurun_out/r3_check2/pytest_fuzz.log 2>&1; tail -3 gpurun_out/r3_check2/pytest_fuzz.log
for args in "" "--one-pass"; do
python bench.py --steps 5 --warmup 2 --no-pmc --no-cpu-baseline --no-online $args 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$args',d['value'],d['ms_per_step'],d['phases_ms'])" | tee -a gpurun_out/r3_check2/bench.txt
done
BENCH_ARGS="--one-pass" bash scripts/jobs/r3_trace.sh
cp gpurun_out/r3_trace/kernel_trace_stats.txt gpurun_out/r3_check2/
