# binary16 mean table after the branch-free accumulation: parity tests, then config 5 at full size
timeout 900 python -m pytest tests/test_gpu_som_kernels.py tests/test_gpu_fuzz_parity.py -m gpu -x -q -k "sums or fuzz or cluster" 2>&1 | tail -3
PXSOM_FUZZ_DTYPE=f16 PXSOM_FUZZ_CASES=300 timeout 900 python -m pytest tests/test_gpu_fuzz_parity.py -m gpu -x -q -k "assign_and_sums or arrow or front" 2>&1 | tail -3
python bench.py --config cfg5 --steps 3 --warmup 1 > gpurun_out/bench_cfg5.json 2> gpurun_out/bench_cfg5.err; python -c "
import json; d=json.load(open('gpurun_out/bench_cfg5.json')); print('cfg5', d['value'], d['ms_per_step'], d['phases_ms'])"
