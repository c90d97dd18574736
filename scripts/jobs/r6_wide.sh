# Round 6: the wide one-launch step with 128 rows per workgroup and one atomic per (run of equal labels, channel): parity + cfg4 lines
# against a variant build (VARIANT=name of ark_analysis_amd/variants/<name>.so)
R=$GRAFT_REPO_ROOT
cd $R; O=gpurun_out/r6_wide; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_schedule.py tests/test_gpu_wide_rows.py tests/test_gpu_alternate_routes.py tests/test_gpu_distributed.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
line() { python scripts/dev/ab_line.py $1 --config cfg4 --steps 3 --warmup 1 --no-cpu-baseline --no-online --no-pmc --no-operating-range 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$1', d['value'], d['ms_per_step'], d['phases_ms']['train_batch'], d['phases_ms']['assign_and_mean_table'])"; }
for r in 1 2 3; do [ -n "$VARIANT" ] && line ark_analysis_amd/variants/$VARIANT.so; line ark_analysis_amd/libpxsom.so; done | tee $O/lines.txt
