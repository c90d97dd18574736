# What does the libm-free gain cost the training pass?  Same box: shipped build against a timing build with -expm1(den log q)
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/r5_gain_cost; mkdir -p $O
line() { python bench.py --no-cpu-baseline --no-online --no-pmc --no-operating-range 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$1', d['value'], d['ms_per_step'], d['phases_ms']['train_batch'], d['phases_ms']['assign_filter_kernel'])"; }
variant() {
  PXSOM_HIPCC_EXTRA="$2" python -c "from ark_analysis_amd import _build; _build.build(force=True)" > $O/build_$1.log 2>&1 || { echo "build $1 failed"; tail -5 $O/build_$1.log; return; }
  for i in 1 2 3; do PXSOM_HIPCC_EXTRA="$2" line "$1"; done
}
{ variant shipped ""; variant expm1_timing "-DPXSOM_GAIN_TIMING_EXPM1"; variant shipped_again ""; } 2>&1 | tee $O/summary.txt
