# Round 5: timing builds of the three-waves-per-SIMD one-pass kernel (what bounds it?): rebuilt on the box per variant.
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r5_onepass_abl; mkdir -p $O
line() { python bench.py --no-cpu-baseline --no-online --no-pmc --no-operating-range 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$1', d['value'], d['phases_ms']['assign_filter_kernel'], d['phases_ms']['train_batch'])"; }
variant() {
  PXSOM_HIPCC_EXTRA="$2" python -c "from ark_analysis_amd import _build; _build.build(force=True)" > $O/build_$1.log 2>&1 || { echo "build $1 failed"; tail -5 $O/build_$1.log; return; }
  PXSOM_HIPCC_EXTRA="$2" line "$1" ; PXSOM_HIPCC_EXTRA="$2" line "$1"
}
{
PXSOM_ONEPASS=0 variant old_kernel_umul24 ""
variant new_768x3 ""
variant new_768x3_noadds "-DPXSOM_ONE_ABL=1"
variant new_768x3_noadds_notop2 "-DPXSOM_ONE_ABL=3"
variant new_512x4 "-DPXSOM_ONE_THREADS=512 -DPXSOM_ONE_WPE=4"
variant new_512x4_noadds "-DPXSOM_ONE_THREADS=512 -DPXSOM_ONE_WPE=4 -DPXSOM_ONE_ABL=1"
variant new_1024x4 "-DPXSOM_ONE_THREADS=1024 -DPXSOM_ONE_WPE=4"
variant new_512x2_256regs "-DPXSOM_ONE_THREADS=512 -DPXSOM_ONE_WPE=2"
} 2>&1 | tee $O/summary.txt
