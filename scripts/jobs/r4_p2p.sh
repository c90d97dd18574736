cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_exchange.py tests/test_gpu_bench_multirank.py tests/test_gpu_distributed.py -q -x > gpurun_out/r4_p2p.log 2>&1; tail -12 gpurun_out/r4_p2p.log | grep -v "RCCL\|HIP version\|ROCm\|Hostname\|Librccl"
