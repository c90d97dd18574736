# cell tables of realistic size (20 K / 50 K / 200 K cells x 100 features) with and without the one-launch wide step -> gpurun_out/r4_wide_small.txt
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
for w in 0 1; do PXSOM_STEP_WIDE=$w python scripts/debug/small_cell_table_probe.py 2>&1 | grep cells; done | tee gpurun_out/r4_wide_small.txt
