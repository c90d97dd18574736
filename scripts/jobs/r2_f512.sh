PXSOM_FUZZ_CASES=800 timeout 1500 python -m pytest tests/test_gpu_fuzz_parity.py -x -q -k "assign or batch" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_som_kernels.py -x -q 2>&1 | tail -2
for c in cfg5 cfg4; do
timeout 900 python bench.py --config $c --steps 3 --warmup 1 --no-pmc --no-cpu-baseline --no-online > gpurun_out/bench_$c.json 2> gpurun_out/bench_$c.err; python -c "
import json; d=json.load(open('gpurun_out/bench_$c.json')); print('$c', d['value'], d['ms_per_step'], d['phases_ms'])"
done
