R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/r6_dbg
PYTHONPATH=$R python $SCRIPT 2>&1 | tail -40 | tee gpurun_out/r6_dbg/out.txt
