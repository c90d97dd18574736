# Round 5: scan path of the one-pass kernel decided once per 64 rows: parity (new coherent-rows test), probe, default bench line against the build without it
R=$GRAFT_REPO_ROOT
cd $R; O=gpurun_out/r5_scan; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_som_kernels.py tests/test_gpu_fuzz_parity.py -m gpu -x -q > $O/pytest2.log 2>&1; tail -3 $O/pytest2.log
python scripts/debug/label_coherence_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/probe_trip52.txt
line() { PXSOM_HIPCC_EXTRA="$2" python bench.py --no-cpu-baseline --no-online --no-pmc --no-operating-range 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$1 cfg2', d['value'], d['ms_per_step'], d['phases_ms'])"; }
{ line scan ""; line scan ""
PXSOM_HIPCC_EXTRA="-DPXSOM_ADD_SCAN=0" python -c "from ark_analysis_amd import _build; _build.build(force=True)" > $O/build_off.log 2>&1 || tail -5 $O/build_off.log
line off "-DPXSOM_ADD_SCAN=0"; line off "-DPXSOM_ADD_SCAN=0"
python -c "from ark_analysis_amd import _build; _build.build(force=True)" > $O/build_on.log 2>&1
line scan ""; line scan ""; } | tee $O/bench_ab.txt
