# Round-5 refresh: whole GPU suite + smoke + default bench, round profile (kernel trace + PMC passes), one bench line per
# other config, the binary64 probe -> gpurun_out/ (summaries are copied into profiles/r05/ afterwards)
R=$GRAFT_REPO_ROOT
cd $R
bash scripts/jobs/all_gpu.sh
bash scripts/jobs/prof_round.sh
mkdir -p gpurun_out/r5_cfgs
for cfg in cfg3 cfg4 cfg5; do
  python bench.py --config $cfg --steps 3 --warmup 1 > gpurun_out/r5_cfgs/bench_$cfg.json 2>/dev/null
  python -c "
import json;d=json.loads(open('gpurun_out/r5_cfgs/bench_$cfg.json').read().strip().splitlines()[-1]);print('$cfg',d['value'],d['ms_per_step'],d['phases_ms'])"
done
python bench.py --two-pass --steps 5 --warmup 2 --no-cpu-baseline --no-online --no-operating-range > gpurun_out/r5_cfgs/bench_cfg2_two_pass.json 2>/dev/null
PXSOM_ONEPASS=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-online --no-operating-range --no-pmc > gpurun_out/r5_cfgs/bench_cfg2_onepass_three_waves.json 2>/dev/null
python -c "
import json
for f in ('bench_cfg2_two_pass','bench_cfg2_onepass_three_waves'):
    d=json.loads(open('gpurun_out/r5_cfgs/%s.json'%f).read().strip().splitlines()[-1]);print(f,d['value'],d['ms_per_step'],d['phases_ms'])"
python scripts/debug/f64_assign_probe.py 2>/dev/null | tee gpurun_out/r5_cfgs/f64_assign_probe.txt
python scripts/debug/f64_train_probe.py 2>/dev/null | tail -6 | tee gpurun_out/r5_cfgs/f64_train_probe.txt
