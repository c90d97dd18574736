# same-box A/B on config 5's shape (8 FOVs of 2048^2 x 40 binary16, 20 x 20 SOM) + parity of the packed-K routes on the default build
R=$GRAFT_REPO_ROOT
cd $R; O=gpurun_out/r6_ab; mkdir -p $O
line() { python scripts/dev/ab_line.py ark_analysis_amd/variants/$1.so --config cfg5 --fovs-per-gpu 8 --steps 3 --warmup 1 --no-cpu-baseline --no-online --no-pmc --no-operating-range 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$1', d['value'], d['ms_per_step'], d['phases_ms'])"; }
for r in $(seq ${REPS:-3}); do for v in $VARIANTS; do line $v; done; done | tee $O/bench_cfg5_$(echo $VARIANTS | tr ' ' '_').txt
timeout 1700 python -m pytest tests/test_gpu_alternate_routes.py tests/test_gpu_som_kernels.py tests/test_gpu_schedule.py -m gpu -x -q 2>&1 | tail -2
PXSOM_FUZZ_DTYPE=f16 PXSOM_FUZZ_CASES=300 timeout 900 python -m pytest tests/test_gpu_fuzz_parity.py -m gpu -x -q 2>&1 | tail -2
