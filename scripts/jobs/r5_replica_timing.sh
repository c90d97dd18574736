# Round 5, timing build only (-DPXSOM_TIMING_REPLICA: wrong results): what would two table replicas by row parity buy the one-pass kernel?
R=$GRAFT_REPO_ROOT
cd $R; O=gpurun_out/r5_replica; mkdir -p $O
line() { PXSOM_HIPCC_EXTRA="$2" python bench.py --no-cpu-baseline --no-online --no-pmc --no-operating-range 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$1 cfg2', d['value'], d['ms_per_step'], d['phases_ms'])"; }
{ line head ""; line head ""
PXSOM_HIPCC_EXTRA="-DPXSOM_TIMING_REPLICA=1" python -c "from ark_analysis_amd import _build; _build.build(force=True)" > $O/build.log 2>&1 || tail -5 $O/build.log
line two_replicas "-DPXSOM_TIMING_REPLICA=1"; line two_replicas "-DPXSOM_TIMING_REPLICA=1"
python -c "from ark_analysis_amd import _build; _build.build(force=True)" > $O/build_on.log 2>&1
line head ""; } | tee $O/summary.txt
