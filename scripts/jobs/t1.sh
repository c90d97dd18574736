timeout 600 python -m pytest tests/test_gpu_som_kernels.py -m gpu -x -q -k "online" 2>&1 | grep -v "^$" | tail -8
