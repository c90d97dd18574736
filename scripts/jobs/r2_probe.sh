python scripts/debug/fused_k8_probe.py 2>&1 | grep "accumulating\|equal\|assign +"
timeout 600 python -m pytest tests/test_gpu_som_kernels.py -x -q -k "assign_sums" 2>&1 | tail -3
