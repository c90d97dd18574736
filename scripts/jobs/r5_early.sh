# Round 5: the trip's scalar decisions formed early (PXSOM_TRIP_EARLY) against where they are used: parity, then interleaved default lines
R=$GRAFT_REPO_ROOT
cd $R; O=gpurun_out/r5_early; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_som_kernels.py tests/test_gpu_fuzz_parity.py tests/test_gpu_schedule.py -m gpu -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
line() { PXSOM_HIPCC_EXTRA="$2" python bench.py --no-cpu-baseline --no-online --no-pmc --no-operating-range 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$1 cfg2', d['value'], d['ms_per_step'], d['phases_ms'])"; }
{ line early ""; line early ""
PXSOM_HIPCC_EXTRA="-DPXSOM_TRIP_EARLY=0" python -c "from ark_analysis_amd import _build; _build.build(force=True)" > $O/build_off.log 2>&1 || tail -5 $O/build_off.log
line late "-DPXSOM_TRIP_EARLY=0"; line late "-DPXSOM_TRIP_EARLY=0"
python -c "from ark_analysis_amd import _build; _build.build(force=True)" > $O/build_on.log 2>&1
line early ""; line early ""; } | tee $O/bench_ab.txt
