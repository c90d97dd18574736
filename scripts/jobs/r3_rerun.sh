R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/all_gpu gpurun_out/r3_cfgs
timeout 3000 python -m pytest tests -m gpu -q > gpurun_out/all_gpu/pytest.log 2>&1; grep -E "passed|failed" gpurun_out/all_gpu/pytest.log | tail -3
for cfg in cfg3 cfg4 cfg5; do
  python bench.py --config $cfg --steps 3 --warmup 1 > gpurun_out/r3_cfgs/bench_$cfg.json 2>/dev/null
  python -c "
import json;d=json.loads(open('gpurun_out/r3_cfgs/bench_$cfg.json').read().strip().splitlines()[-1]);print('$cfg',d['value'],d['ms_per_step'],d['phases_ms'],d.get('cpu_baseline'))"
done
bash scripts/jobs/prof_round.sh
