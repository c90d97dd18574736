# Round-3 refresh: whole GPU suite + smoke + default bench, round profile (kernel trace + PMC passes), pre-processing
# profile, one bench line per other config -> gpurun_out/ (summaries are copied into profiles/r03/ afterwards)
R=$GRAFT_REPO_ROOT
cd $R
bash scripts/jobs/all_gpu.sh
bash scripts/jobs/prof_round.sh
bash scripts/jobs/r3_pre_prof.sh
mkdir -p gpurun_out/r3_cfgs
for cfg in cfg3 cfg4 cfg5; do
  python bench.py --config $cfg --steps 3 --warmup 1 > gpurun_out/r3_cfgs/bench_$cfg.json 2>/dev/null
  python -c "
import json;d=json.loads(open('gpurun_out/r3_cfgs/bench_$cfg.json').read().strip().splitlines()[-1]);print('$cfg',d['value'],d['ms_per_step'],d['phases_ms'])"
done
python bench.py --two-pass --steps 5 --warmup 2 --no-cpu-baseline --no-online --no-operating-range > gpurun_out/r3_cfgs/bench_cfg2_two_pass.json 2>/dev/null
python bench.py --batch-steps 64 --steps 5 --warmup 2 --no-cpu-baseline --no-online --no-operating-range --no-pmc > gpurun_out/r3_cfgs/bench_cfg2_equal64.json 2>/dev/null
