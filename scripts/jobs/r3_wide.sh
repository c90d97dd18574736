R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r3_wide
cd $R
timeout 900 python -m pytest tests/test_gpu_wide_rows.py -x -q > gpurun_out/r3_wide/pytest.log 2>&1; tail -15 gpurun_out/r3_wide/pytest.log
bash scripts/jobs/all_gpu.sh
