# Round 6: cross terms of the streamed filter / wide step in an accumulator of their own (tighter tolerance): parity + cfg4 / cfg5 lines
R=$GRAFT_REPO_ROOT
cd $R; O=gpurun_out/r6_split; mkdir -p $O
timeout 2000 python -m pytest tests/test_gpu_schedule.py tests/test_gpu_wide_rows.py tests/test_gpu_alternate_routes.py tests/test_gpu_som_kernels.py tests/test_gpu_fuzz_parity.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
line() { python scripts/dev/ab_line.py $1 --config $2 --steps 3 --warmup 1 --no-cpu-baseline --no-online --no-pmc --no-operating-range 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$1 $2', d['value'], d['ms_per_step'], d['phases_ms']['train_batch'], d['phases_ms']['assign_and_mean_table'], d['phases_ms']['assign_filter_kernel'], d['phases_ms']['assign_exact_rows'])"; }
for r in 1 2 3; do line ark_analysis_amd/variants/nosplit.so cfg4; line ark_analysis_amd/libpxsom.so cfg4; done | tee $O/lines.txt
