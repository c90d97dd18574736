python scripts/debug/k8_cost.py 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_som_kernels.py -m gpu -x -q -k "cluster_sums or full_size or accumulate" 2>&1 | tail -2
