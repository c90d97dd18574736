# Round 6: the default line with the phase events inside the timed region (HEAD~: bench_old_tmp.py) and outside it, interleaved
R=$GRAFT_REPO_ROOT
cd $R; O=gpurun_out/r6_benchab; mkdir -p $O
line() { python $1 --no-cpu-baseline --no-online --no-pmc --no-operating-range 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$1', d['value'], d['ms_per_step'], d['phases_ms']['train_batch'], d['phases_ms']['assign_and_mean_table'], d['phases_ms']['assign_filter_kernel'], d['roofline']['frac'], d['roofline_step']['frac'])"; }
for r in 1 2 3 4; do line bench_old_tmp.py; line bench.py; done | tee $O/lines.txt
