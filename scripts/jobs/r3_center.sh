# centred filter: parity (assign, one-pass, training, fuzz, pipeline), then the bench line with the operating range + kernel trace
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r3_center; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_som_kernels.py tests/test_gpu_fuzz_parity.py tests/test_gpu_schedule.py tests/test_pipeline_dropin.py tests/test_gpu_exchange.py tests/test_gpu_bench_multirank.py -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for i in 1 2; do python bench.py --steps 10 --warmup 2 --no-pmc --no-cpu-baseline --no-online --no-operating-range 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['phases_ms'])" | tee -a $O/bench.txt; done
bash scripts/jobs/r3_trace.sh > /dev/null 2>&1
grep "batch_step_kernel" gpurun_out/r3_trace/kernel_trace_stats.txt | tail -8 | cut -c1-60,95-180
PXSOM_FUZZ_CASES=600 PXSOM_FUZZ_SEED=4242 timeout 1500 python -m pytest tests/test_gpu_fuzz_parity.py -q -x 2>&1 | tail -2
