timeout 900 python -m pytest tests/test_gpu_som_kernels.py -m gpu -x -q -k "cluster_sums or batch_accumulate or update_prepare" 2>&1 | tail -5
python scripts/debug/sums_check.py
