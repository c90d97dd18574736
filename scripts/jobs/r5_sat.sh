# batch_gain with the saturation shortcut: parity of every batch-training test, the crowded whole-run sweep, three bench lines
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/r5_sat; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_schedule.py tests/test_gpu_fuzz_parity.py tests/test_pipeline_dropin.py tests/test_gpu_som_kernels.py tests/test_gpu_wide_rows.py tests/test_gpu_exchange.py tests/test_gpu_distributed.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
PXSOM_FUZZ_CASES=300 timeout 1500 python -m pytest tests/test_gpu_fuzz_parity.py -m gpu -x -q -k "batch" > $O/fuzz300.log 2>&1; tail -2 $O/fuzz300.log
for i in 1 2 3; do python bench.py --no-cpu-baseline --no-online --no-pmc --no-operating-range 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('saturation', d['value'], d['ms_per_step'], d['phases_ms']['train_batch'], d['phases_ms']['assign_filter_kernel'])"; done | tee $O/summary.txt
