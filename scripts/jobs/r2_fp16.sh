timeout 1200 python -m pytest tests/test_gpu_som_kernels.py -x -q -k "f16 or fp16 or half or config5 or shapes or dtype" 2>&1 | tail -3
timeout 600 python bench.py --config cfg5 --steps 5 --warmup 2 --no-pmc --no-cpu-baseline --no-online > gpurun_out/cfg5.json 2> gpurun_out/cfg5.err; python -c "
import json; d=json.load(open('gpurun_out/cfg5.json')); print(d['value'], d['ms_per_step'], d['phases_ms'])"
