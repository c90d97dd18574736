timeout 900 python -m pytest tests/test_gpu_exchange.py tests/test_gpu_distributed.py -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -8
