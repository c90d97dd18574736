# Round-6 refresh: whole GPU suite + smoke + default bench, round profile (kernel trace + PMC passes), one bench line per other
# config (cfg1 = the plumbing config), two-pass line, pipeline functions, label-coherence probe, phase stamps -> gpurun_out/
# (summaries are copied into profiles/r06/ afterwards)
R=$GRAFT_REPO_ROOT
cd $R
bash scripts/jobs/all_gpu.sh
bash scripts/jobs/prof_round.sh
O=gpurun_out/r6_cfgs; mkdir -p $O
for cfg in cfg3 cfg4 cfg5; do
  python bench.py --config $cfg --steps 3 --warmup 1 > $O/bench_$cfg.json 2>/dev/null
  python -c "
import json;d=json.loads(open('$O/bench_$cfg.json').read().strip().splitlines()[-1]);print('$cfg',d['value'],d['ms_per_step'],d['phases_ms'],d.get('roofline',{}).get('frac'))"
done
python bench.py --config cfg1 --steps 5 --warmup 1 > $O/bench_cfg1.json 2>/dev/null; tail -1 $O/bench_cfg1.json | cut -c1-400
python bench.py --two-pass --steps 5 --warmup 2 --no-cpu-baseline --no-online --no-operating-range > $O/bench_cfg2_two_pass.json 2>/dev/null
python scripts/pipeline_bench.py --fovs 40 > $O/cluster_pixels.json 2>/dev/null; tail -1 $O/cluster_pixels.json
python scripts/preprocess_bench.py --fovs 30 > $O/create_pixel_matrix.json 2>/dev/null; tail -1 $O/create_pixel_matrix.json
python scripts/debug/label_coherence_probe.py 2>/dev/null | tee $O/label_coherence.txt | tail -8
python scripts/debug/f64_assign_probe.py 2>/dev/null | tee $O/f64_assign_probe.txt | tail -4
bash scripts/jobs/r6_phase.sh > /dev/null 2>&1
BENCH_ARGS="--config cfg4" N=80 bash scripts/jobs/r6_timeline.sh > /dev/null 2>&1
