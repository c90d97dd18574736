# screened exact kernel: parity (kernel tests + wide rows + randomised sweep), config 4 bench line + kernel trace -> gpurun_out/r4_exact/
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r4_exact
cd $R
timeout 2400 python -m pytest tests/test_gpu_som_kernels.py tests/test_gpu_fuzz_parity.py tests/test_gpu_wide_rows.py tests/test_gpu_schedule.py -m gpu -q > gpurun_out/r4_exact/pytest.log 2>&1; tail -6 gpurun_out/r4_exact/pytest.log
cd /tmp && export TMPDIR=/tmp
for c in cfg4 $EXTRA_CFG; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$c -o t -- python $R/bench.py --config $c --steps 3 --warmup 1 --no-pmc --no-cpu-baseline > $R/gpurun_out/r4_exact/bench_$c.json 2>/dev/null
  python $R/scripts/prof_summarize.py /tmp/p_$c $R/gpurun_out/r4_exact/kernel_trace_$c.txt bmu_ cluster_sums batch_ centring stats_ gather_steps > /dev/null
  echo "== $c"; python -c "
import json;d=json.loads(open('$R/gpurun_out/r4_exact/bench_$c.json').read().strip().splitlines()[-1]);print('$c',d['value'],d['ms_per_step'],d['phases_ms'])"
  grep -E "bmu_|cluster_sums|batch_|gather" $R/gpurun_out/r4_exact/kernel_trace_$c.txt | head -14 | cut -c1-70,100-190
done
