# End-of-round refresh: whole GPU suite + smoke + default bench, the round profile, the robustness table, the other configs
bash scripts/jobs/all_gpu.sh
bash scripts/jobs/prof_round.sh
timeout 900 python tests/tools/robustness_sweep.py > gpurun_out/robustness_sweep.txt 2>&1; tail -3 gpurun_out/robustness_sweep.txt | cut -c1-200
for cfg in cfg3 cfg4 cfg5; do python bench.py --config $cfg --steps 3 --warmup 1 > gpurun_out/bench_$cfg.json 2> gpurun_out/bench_$cfg.err; python -c "
import json; d=json.load(open('gpurun_out/bench_$cfg.json')); print('$cfg', d['value'], d['ms_per_step'], d['phases_ms'], d['roofline']['frac'], d.get('mfma_util',{}).get('value'))"; done
cp gpurun_out/bench_default.json gpurun_out/bench_cfg2.json
