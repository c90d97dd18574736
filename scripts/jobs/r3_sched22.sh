# the 6 + 15 + 1x5 default schedule: parity tests, bench with quality against the online rule, kernel trace
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r3_sched22; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_som_kernels.py tests/test_gpu_schedule.py tests/test_gpu_fuzz_parity.py tests/test_pipeline_dropin.py tests/test_gpu_exchange.py tests/test_gpu_bench_multirank.py -x -q > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2
for i in 1 2; do python bench.py --steps 10 --warmup 2 --no-pmc --no-cpu-baseline --no-online --no-operating-range 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['phases_ms'])" | tee -a $O/bench.txt; done
python bench.py --steps 10 --warmup 2 --no-pmc --no-operating-range --no-online > $O/bench_quality.json 2>$O/bench_quality.err
python -c "
import json;d=json.loads(open('$O/bench_quality.json').read().strip().splitlines()[-1]);print(d['value'],d.get('batch_train'))"
bash scripts/jobs/r3_trace.sh > /dev/null 2>&1
grep "batch_step_kernel" gpurun_out/r3_trace/kernel_trace_stats.txt | tail -9 | cut -c1-60,95-180
