R=$GRAFT_REPO_ROOT
cd $R; O=gpurun_out/r6_cfg5; mkdir -p $O
for i in 1 2; do python bench.py --config cfg5 --fovs-per-gpu 8 --steps 3 --warmup 1 --no-cpu-baseline --no-online --no-pmc --no-operating-range 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('head', d['value'], d['ms_per_step'], d['phases_ms'])"; done | tee $O/bench.txt
timeout 1700 python -m pytest tests/test_gpu_alternate_routes.py tests/test_gpu_som_kernels.py tests/test_gpu_schedule.py -m gpu -x -q 2>&1 | tail -2
