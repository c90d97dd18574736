timeout 600 python -m pytest tests/test_gpu_som_kernels.py -m gpu -x -q -k "cell_som_shape" 2>&1 | grep -v "^$" | tail -15
