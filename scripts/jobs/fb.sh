timeout 600 python -m pytest tests/test_gpu_som_kernels.py -m gpu -x -q -k "batch or accumulate" 2>&1 | tail -2
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-online | grep -o '"value": [0-9.]*\|"phases_ms".*"frac": [0-9.]*'
