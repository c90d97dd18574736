# two-stage filter: parity, bench, operating range
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r3_two; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_som_kernels.py tests/test_gpu_fuzz_parity.py tests/test_pipeline_dropin.py -x -q > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2
python bench.py --steps 10 --warmup 2 --no-pmc --no-cpu-baseline --no-online 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['phases_ms']);print(json.dumps(d['operating_range']))" | tee -a $O/bench.txt
python bench.py --two-pass --steps 10 --warmup 2 --no-pmc --no-cpu-baseline --no-online --no-operating-range 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['phases_ms'])" | tee -a $O/bench.txt
