# step kernel changes: parity (schedule tests, som kernels, fuzz quick), bench default + kernel trace -> gpurun_out/r4_step/
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r4_step
cd $R
timeout 2400 python -m pytest tests/test_gpu_schedule.py tests/test_gpu_som_kernels.py tests/test_gpu_fuzz_parity.py -m gpu -q > gpurun_out/r4_step/pytest.log 2>&1; tail -8 gpurun_out/r4_step/pytest.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-online --no-pmc --no-operating-range > gpurun_out/r4_step/bench.json 2> gpurun_out/r4_step/bench.err
python -c "
import json;d=json.loads(open('gpurun_out/r4_step/bench.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['phases_ms'],d.get('batch_train'))"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o trace -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-online --no-pmc --no-operating-range > /dev/null 2>&1
python $R/scripts/prof_summarize.py /tmp/tr $R/gpurun_out/r4_step/kernel_trace_stats.txt > /dev/null
grep -E "batch_step_kernel|bmu_filter_fast" $R/gpurun_out/r4_step/kernel_trace_stats.txt | cut -c1-60,90-200
