# screened exact kernel: focused parity tests, the fuzz sweep, the robustness table, configs 4 and 5
timeout 900 python -m pytest tests/test_gpu_som_kernels.py -m gpu -x -q 2>&1 | tail -5
PXSOM_FUZZ_CASES=150 timeout 1200 python -m pytest tests/test_gpu_fuzz_parity.py -m gpu -x -q 2>&1 | tail -3
PXSOM_SCREEN_MIN_ROWS=1 PXSOM_FUZZ_CASES=300 timeout 1200 python -m pytest tests/test_gpu_fuzz_parity.py -m gpu -x -q 2>&1 | tail -3
timeout 600 python tests/tools/robustness_sweep.py 2>&1 | tail -12
for cfg in cfg4 cfg5; do python bench.py --config $cfg --steps 3 --warmup 1 > gpurun_out/bench_$cfg.json 2> gpurun_out/bench_$cfg.err; python -c "
import json; d=json.load(open('gpurun_out/bench_$cfg.json')); print('$cfg', d['value'], d['ms_per_step'], d['phases_ms'])"; done
