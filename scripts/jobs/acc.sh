timeout 600 python -m pytest tests/test_gpu_som_kernels.py tests/test_gpu_distributed.py -m gpu -x -q -k "batch or distributed or rank" 2>&1 | tail -15
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-online | grep -o '"value": [0-9.]*\|"phases_ms".*"frac": [0-9.]*'
