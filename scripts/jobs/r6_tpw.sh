# Round 6: tail steps with one / two tiles per wave (fewer workgroups flush fewer atomics), default bench lines
R=$GRAFT_REPO_ROOT
cd $R; O=gpurun_out/r6_tpw; mkdir -p $O
line() { python bench.py --no-cpu-baseline --no-online --no-pmc --no-operating-range 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$1', d['value'], d['ms_per_step'], d['phases_ms'])"; }
{ line tpw1; PXSOM_STEP_TPW_SMALL=2 line tpw2; line tpw1; PXSOM_STEP_TPW_SMALL=2 line tpw2; line tpw1; } | tee $O/bench.txt
