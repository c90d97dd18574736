R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r3_cfg45
timeout 600 python -m pytest tests/test_gpu_exchange.py -x -q > gpurun_out/r3_cfg45/pytest_exchange.log 2>&1; tail -3 gpurun_out/r3_cfg45/pytest_exchange.log
bash scripts/jobs/r3_cfg45.sh
