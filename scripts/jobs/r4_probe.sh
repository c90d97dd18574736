cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python scripts/debug/assign_sums_size_probe.py 2>&1 | grep rows | tee gpurun_out/r4_size_probe.txt
