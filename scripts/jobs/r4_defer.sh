# one-pass kernel with the deferred full search: kernel parity tests + sweep, default bench with the operating range -> gpurun_out/r4_defer/
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r4_defer
timeout 2400 python -m pytest tests/test_gpu_som_kernels.py tests/test_gpu_fuzz_parity.py tests/test_gpu_schedule.py -m gpu -q > gpurun_out/r4_defer/pytest.log 2>&1; tail -3 gpurun_out/r4_defer/pytest.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-online --no-pmc > gpurun_out/r4_defer/bench.json 2>/dev/null
python -c "
import json;d=json.loads(open('gpurun_out/r4_defer/bench.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['phases_ms'],d['roofline']['frac']);print({k:(v['assign_ms'],v['labels_and_mean_table_one_pass_ms']) for k,v in d['operating_range'].items() if isinstance(v,dict)})"
