# centred streamed filter: parity (kernel tests + randomised sweep), config 4 / 5 bench lines -> gpurun_out/r4_centre/
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r4_centre
cd $R
timeout 2400 python -m pytest tests/test_gpu_som_kernels.py tests/test_gpu_fuzz_parity.py tests/test_gpu_wide_rows.py tests/test_gpu_schedule.py -m gpu -q > gpurun_out/r4_centre/pytest.log 2>&1; tail -6 gpurun_out/r4_centre/pytest.log
for cfg in cfg4 cfg5; do
  timeout 900 python bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-pmc > gpurun_out/r4_centre/bench_$cfg.json 2>/dev/null
  python -c "
import json;d=json.loads(open('gpurun_out/r4_centre/bench_$cfg.json').read().strip().splitlines()[-1]);print('$cfg',d['value'],d['ms_per_step'],d['phases_ms'])"
done
