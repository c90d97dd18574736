R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/r6_size
python scripts/debug/assign_sums_size_probe.py 2>/dev/null | tee gpurun_out/r6_size/size_probe.txt
