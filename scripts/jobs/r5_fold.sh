# Round 5: count adds folded into the idle lanes of the last lane group (24 instead of 28 LDS atomics per 64 rows): whole GPU suite, then
# same-box A/B against round 4's tree (unpacked under _r4ref/ for this run only) on configs 2 and 3
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r5_fold; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
line() { python bench.py $2 --no-cpu-baseline --no-online --no-pmc --no-operating-range 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$1', d['value'], d['ms_per_step'], d['phases_ms']['train_batch'], d['phases_ms']['assign_filter_kernel'])"; }
{
for i in 1 2; do line "HEAD cfg2" ""; done
line "HEAD cfg3" "--config cfg3 --steps 5 --warmup 2"
if [ -d _r4ref ]; then cd _r4ref; python -c "from ark_analysis_amd import _build; _build.build()" > $O/build_r4.log 2>&1
for i in 1 2; do line "round4 cfg2" ""; done
line "round4 cfg3" "--config cfg3 --steps 5 --warmup 2"; cd $R; fi
for i in 1 2; do line "HEAD cfg2" ""; done
line "HEAD cfg3" "--config cfg3 --steps 5 --warmup 2"
} 2>&1 | tee $O/summary.txt
