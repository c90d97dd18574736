python scripts/debug/online_shapes.py
timeout 600 python -m pytest tests/test_gpu_som_kernels.py -m gpu -x -q -k "online" 2>&1 | tail -2
