R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r3_robust
python -m tests.tools.robustness_sweep > gpurun_out/r3_robust/robustness_sweep.txt 2>&1; tail -4 gpurun_out/r3_robust/robustness_sweep.txt | cut -c1-250
python bench.py --steps 5 --warmup 2 --no-pmc --no-cpu-baseline --no-online 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(json.dumps(d['operating_range'],indent=1))" | tee gpurun_out/r3_robust/operating_range.json
