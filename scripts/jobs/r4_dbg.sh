cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; python scripts/debug/tail_case_probe.py 2>&1 | tee gpurun_out/r4_dbg.txt | tail -40
