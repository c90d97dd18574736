# Round 6: same-box A/B of variant builds (ark_analysis_amd/variants/*.so), interleaved default bench lines.  VARIANTS="a b" REPS=3
R=$GRAFT_REPO_ROOT
cd $R; O=gpurun_out/r6_ab; mkdir -p $O
line() { python scripts/dev/ab_line.py ark_analysis_amd/variants/$1.so --no-cpu-baseline --no-online --no-pmc --no-operating-range $BENCH_ARGS 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$1', d['value'], d['ms_per_step'], d['phases_ms'])"; }
for r in $(seq ${REPS:-3}); do for v in $VARIANTS; do line $v; done; done | tee $O/bench_$(echo $VARIANTS | tr ' ' '_').txt
