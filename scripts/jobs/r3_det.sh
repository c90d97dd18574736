R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r3_det
cd $R
timeout 900 python -m pytest tests/test_gpu_schedule.py tests/test_gpu_wide_rows.py -x -q > gpurun_out/r3_det/pytest.log 2>&1; tail -15 gpurun_out/r3_det/pytest.log
timeout 900 python -m pytest tests/test_pipeline_dropin.py tests/test_gpu_som_kernels.py tests/test_gpu_distributed.py tests/test_gpu_exchange.py -m gpu -x -q > gpurun_out/r3_det/pytest2.log 2>&1; tail -5 gpurun_out/r3_det/pytest2.log
