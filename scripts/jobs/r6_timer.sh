# Round 6: kernel timer on the dispatch's own events (hipExtLaunchKernelGGL): bench lines + a few other configs + rocprof agreement
R=$GRAFT_REPO_ROOT
cd $R; O=gpurun_out/r6_timer; mkdir -p $O
line() { python bench.py $2 --no-cpu-baseline --no-online --no-pmc --no-operating-range 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$1', d['value'], d['ms_per_step'], d['phases_ms']['train_batch'], d['phases_ms']['assign_and_mean_table'], d['phases_ms']['assign_filter_kernel'], d['roofline']['frac'], d['roofline'].get('launches_timed'), d['roofline_step']['frac'])"; }
{ line cfg2; line cfg2; line cfg2; line cfg3 "--config cfg3 --steps 3 --warmup 1"; line cfg4 "--config cfg4 --steps 3 --warmup 1"; line cfg5 "--config cfg5 --steps 3 --warmup 1"; line twopass "--two-pass"; } 2>&1 | tee $O/lines.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o t -- python $R/bench.py --no-cpu-baseline --no-online --no-pmc --no-operating-range > $R/$O/under_rocprof.log 2>&1
cd $R
grep "^{" $O/under_rocprof.log | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('under rocprof', d['ms_per_step'], d['phases_ms']['assign_filter_kernel'])"
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && grep -E "bmu_filter_fast" "$f" | cut -c1-60,150-400 | head -3
