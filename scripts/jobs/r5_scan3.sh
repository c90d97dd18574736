# scan path out of line: default bench line against the build without it, interleaved
R=$GRAFT_REPO_ROOT
cd $R; O=gpurun_out/r5_scan; mkdir -p $O
line() { PXSOM_HIPCC_EXTRA="$2" python bench.py --no-cpu-baseline --no-online --no-pmc --no-operating-range 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$1 cfg2', d['value'], d['ms_per_step'], d['phases_ms'])"; }
{ line scan ""; line scan ""
PXSOM_HIPCC_EXTRA="-DPXSOM_ADD_SCAN=0" python -c "from ark_analysis_amd import _build; _build.build(force=True)" > $O/build_off.log 2>&1 || tail -5 $O/build_off.log
line off "-DPXSOM_ADD_SCAN=0"; line off "-DPXSOM_ADD_SCAN=0"
python -c "from ark_analysis_amd import _build; _build.build(force=True)" > $O/build_on.log 2>&1
line scan ""; python scripts/debug/label_coherence_probe.py 2>&1 | grep -v amdgpu.ids | head -3; } | tee $O/bench_ab3.txt
