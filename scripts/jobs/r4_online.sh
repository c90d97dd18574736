cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_som_kernels.py tests/test_pyflowsom_vectors.py tests/test_abi_and_host.py -q -k "online or abi or recalled or product" > gpurun_out/r4_online.log 2>&1; tail -6 gpurun_out/r4_online.log
