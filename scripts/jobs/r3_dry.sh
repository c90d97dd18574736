R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r3_dry
timeout 1500 python -m pytest tests/test_gpu_bench_multirank.py tests/test_gpu_distributed.py tests/test_gpu_exchange.py tests/test_pipeline_multirank.py -m gpu -x -q > gpurun_out/r3_dry/pytest.log 2>&1; grep -E "passed|failed|Error|assert" gpurun_out/r3_dry/pytest.log | tail -12
