# screened exact kernel with node groups: parity, then config 4 at several list-length crossovers -> gpurun_out/r4_exact/
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r4_exact
cd $R
timeout 2400 python -m pytest tests/test_gpu_som_kernels.py tests/test_gpu_fuzz_parity.py tests/test_gpu_wide_rows.py tests/test_gpu_schedule.py -m gpu -q > gpurun_out/r4_exact/pytest.log 2>&1; tail -4 gpurun_out/r4_exact/pytest.log
for m in default 512 2048 8192; do
  if [ $m = default ]; then unset PXSOM_SCREEN_MIN_ROWS; else export PXSOM_SCREEN_MIN_ROWS=$m; fi
  timeout 900 python bench.py --config cfg4 --steps 3 --warmup 1 --no-cpu-baseline --no-pmc > gpurun_out/r4_exact/bench_cfg4_$m.json 2>/dev/null
  python -c "
import json;d=json.loads(open('gpurun_out/r4_exact/bench_cfg4_$m.json').read().strip().splitlines()[-1]);print('min_rows $m',d['value'],d['ms_per_step'],d['phases_ms'])"
done
unset PXSOM_SCREEN_MIN_ROWS
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p4 -o t -- python $R/bench.py --config cfg4 --steps 3 --warmup 1 --no-pmc --no-cpu-baseline > /dev/null 2>&1
python $R/scripts/prof_summarize.py /tmp/p4 $R/gpurun_out/r4_exact/kernel_trace_cfg4.txt bmu_ cluster_sums batch_ centring stats_ gather_steps > /dev/null
grep -E "bmu_|cluster_sums|batch_|gather" $R/gpurun_out/r4_exact/kernel_trace_cfg4.txt | head -9 | cut -c1-70,100-190
