# odd table strides (one-pass kernel + fused step): parity + default bench + kernel trace -> gpurun_out/r4_acc/
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r4_acc
cd $R
timeout 2400 python -m pytest tests/test_gpu_som_kernels.py tests/test_gpu_fuzz_parity.py tests/test_gpu_schedule.py -m gpu -q > gpurun_out/r4_acc/pytest.log 2>&1; tail -3 gpurun_out/r4_acc/pytest.log
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-online --no-pmc --no-operating-range > gpurun_out/r4_acc/bench$i.json 2> gpurun_out/r4_acc/bench.err
python -c "
import json;d=json.loads(open('gpurun_out/r4_acc/bench$i.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['phases_ms'])"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o trace -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-online --no-pmc --no-operating-range > /dev/null 2>&1
python $R/scripts/prof_summarize.py /tmp/tr $R/gpurun_out/r4_acc/kernel_trace_stats.txt > /dev/null
grep -E "batch_step_kernel|bmu_filter_fast" $R/gpurun_out/r4_acc/kernel_trace_stats.txt | grep -v "StepArg " | cut -c1-40,96-200
