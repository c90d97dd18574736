# Round 5: the winner's node id formed branch-free (XOR under a select) against the two-armed expression the compiler made a divergent if / else of
R=$GRAFT_REPO_ROOT
cd $R; O=gpurun_out/r5_nodeid; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_som_kernels.py tests/test_gpu_fuzz_parity.py -m gpu -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
line() { PXSOM_HIPCC_EXTRA="$2" python bench.py --no-cpu-baseline --no-online --no-pmc --no-operating-range 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$1 cfg2', d['value'], d['ms_per_step'], d['phases_ms'])"; }
{ line xor ""; line xor ""
PXSOM_HIPCC_EXTRA="-DPXSOM_NODE_ID_TWO_ARMED=1" python -c "from ark_analysis_amd import _build; _build.build(force=True)" > $O/build_old.log 2>&1 || tail -5 $O/build_old.log
line two_armed "-DPXSOM_NODE_ID_TWO_ARMED=1"; line two_armed "-DPXSOM_NODE_ID_TWO_ARMED=1"
python -c "from ark_analysis_amd import _build; _build.build(force=True)" > $O/build_new.log 2>&1
line xor ""; line xor ""; } | tee $O/bench_ab.txt
