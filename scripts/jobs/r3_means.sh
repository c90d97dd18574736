R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r3_means
timeout 600 python -m pytest tests/test_gpu_som_kernels.py -x -q -k "assign_means or one_pass" > gpurun_out/r3_means/pytest.log 2>&1; tail -3 gpurun_out/r3_means/pytest.log
python bench.py --steps 10 --warmup 2 --no-pmc --no-cpu-baseline --no-online --no-operating-range 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['phases_ms'])" | tee gpurun_out/r3_means/bench.txt
SEEDS="41 42 43 44 45" bash scripts/jobs/r3_fuzz.sh 2500
