cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5_uneven
timeout 1500 python -m pytest tests/test_gpu_exchange.py -m gpu -x -q -k "uneven or fused" > gpurun_out/r5_uneven/pytest.log 2>&1; tail -15 gpurun_out/r5_uneven/pytest.log | cut -c1-200
