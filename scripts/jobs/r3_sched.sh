# Round 3: the scheduled batch rule -- parity tests, then bench lines (default two-phase schedule, 64 equal steps).
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r3_sched
cd $R
timeout 900 python -m pytest tests/test_gpu_schedule.py -x -q > gpurun_out/r3_sched/pytest.log 2>&1; tail -5 gpurun_out/r3_sched/pytest.log
timeout 600 python -m pytest tests/test_gpu_som_kernels.py -x -q -k "batch" > gpurun_out/r3_sched/pytest_batch.log 2>&1; tail -3 gpurun_out/r3_sched/pytest_batch.log
python bench.py --steps 5 --warmup 2 --no-pmc --no-cpu-baseline --no-online > gpurun_out/r3_sched/bench_two_phase.json 2> gpurun_out/r3_sched/bench_two_phase.err; cut -c1-900 gpurun_out/r3_sched/bench_two_phase.json
python bench.py --steps 5 --warmup 2 --no-pmc --no-cpu-baseline --no-online --batch-steps 64 > gpurun_out/r3_sched/bench_equal64.json 2>/dev/null; cut -c1-900 gpurun_out/r3_sched/bench_equal64.json
for cfg in cfg4 cfg5; do python bench.py --config $cfg --steps 3 --warmup 1 --no-pmc > gpurun_out/r3_sched/bench_$cfg.json 2>/dev/null; cut -c1-700 gpurun_out/r3_sched/bench_$cfg.json; done
