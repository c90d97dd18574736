R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r3_packed
timeout 900 python -m pytest tests/test_gpu_som_kernels.py -x -q -k "assign_matches_oracle or packed or batch_train_steps_fused or config5" > gpurun_out/r3_packed/pytest.log 2>&1; tail -5 gpurun_out/r3_packed/pytest.log
timeout 900 python -m pytest tests/test_gpu_schedule.py tests/test_gpu_fuzz_parity.py -x -q > gpurun_out/r3_packed/pytest2.log 2>&1; tail -3 gpurun_out/r3_packed/pytest2.log
python bench.py --config cfg5 --steps 3 --warmup 1 --no-pmc 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('cfg5',d['value'],d['ms_per_step'],d['phases_ms'],d['roofline'])" | tee gpurun_out/r3_packed/cfg5.txt
