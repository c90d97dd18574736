# Round 5: what do the table adds of the one-pass kernel cost, and why?  Timing builds (rebuilt on the box per variant).
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r5_onepass_abl2; mkdir -p $O
line() { python bench.py --no-cpu-baseline --no-online --no-pmc --no-operating-range 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$1', d['value'], d['phases_ms']['assign_filter_kernel'], d['phases_ms']['train_batch'])"; }
variant() {
  PXSOM_HIPCC_EXTRA="$2" python -c "from ark_analysis_amd import _build; _build.build(force=True)" > $O/build_$1.log 2>&1 || { echo "build $1 failed"; tail -5 $O/build_$1.log; return; }
  PXSOM_HIPCC_EXTRA="$2" line "$1" ; PXSOM_HIPCC_EXTRA="$2" line "$1"
}
{
variant new_768x3 ""
variant conflict_free_addresses "-DPXSOM_ONE_ABL=4"
variant values_formed_not_added "-DPXSOM_ONE_ABL=8"
variant half_the_adds "-DPXSOM_ONE_ABL=16"
variant half_the_adds_conflict_free "-DPXSOM_ONE_ABL=20"
} 2>&1 | tee $O/summary.txt
