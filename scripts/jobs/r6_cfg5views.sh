# Round 6: config 5's training steps on row views (no gathered copy): parity + cfg5 lines against the build before
R=$GRAFT_REPO_ROOT
cd $R; O=gpurun_out/r6_cfg5views; mkdir -p $O
timeout 2000 python -m pytest tests/test_gpu_schedule.py tests/test_gpu_wide_rows.py tests/test_gpu_som_kernels.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
line() { python scripts/dev/ab_line.py $1 --config cfg5 --steps 3 --warmup 1 --no-cpu-baseline --no-online --no-pmc --no-operating-range 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$1', d['value'], d['ms_per_step'], d['phases_ms']['train_batch'], d['phases_ms']['assign_and_mean_table'])"; }
for r in 1 2; do line ark_analysis_amd/variants/pre5.so; line ark_analysis_amd/libpxsom.so; done | tee $O/lines.txt
