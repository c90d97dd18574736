# one-launch BMU-only step for wide codebooks (pxsom_batch_step_wide.hip): parity, then config 4 / 5 with and without -> gpurun_out/r4_wide/
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r4_wide
cd $R
timeout 2400 python -m pytest tests/test_gpu_som_kernels.py tests/test_gpu_fuzz_parity.py tests/test_gpu_wide_rows.py tests/test_gpu_schedule.py tests/test_gpu_exchange.py -m gpu -q -x > gpurun_out/r4_wide/pytest.log 2>&1; tail -12 gpurun_out/r4_wide/pytest.log
for w in 0 1; do
  for cfg in cfg4; do
    PXSOM_STEP_WIDE=$w timeout 900 python bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-pmc > gpurun_out/r4_wide/bench_${cfg}_wide$w.json 2>/dev/null
    python -c "
import json;d=json.loads(open('gpurun_out/r4_wide/bench_${cfg}_wide$w.json').read().strip().splitlines()[-1]);print('$cfg wide=$w',d['value'],d['ms_per_step'],d['phases_ms'],d.get('batch_train'))"
  done
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p4 -o t -- python $R/bench.py --config cfg4 --steps 3 --warmup 1 --no-pmc --no-cpu-baseline > /dev/null 2>&1
python $R/scripts/prof_summarize.py /tmp/p4 $R/gpurun_out/r4_wide/kernel_trace_cfg4.txt bmu_ cluster_sums batch_ centring stats_ gather_steps > /dev/null
grep -E "bmu_|cluster_sums|batch_|gather" $R/gpurun_out/r4_wide/kernel_trace_cfg4.txt | head -10 | cut -c1-70,100-190
