# cell tables of realistic size (20 K / 50 K / 200 K cells x 100 features): launch-per-phase route, and the one-launch wide step with its
# windowed steps capped at 0 / 1024 / 4096 / 16384 rows -> gpurun_out/r4_wide_small.txt
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
{ PXSOM_STEP_WIDE=0 python scripts/debug/small_cell_table_probe.py 2>&1 | grep cells
  for cap in 0 1024 4096 16384; do echo "windowed steps up to $cap rows"; PXSOM_STEP_WIDE_WINCAP=$cap python scripts/debug/small_cell_table_probe.py 2>&1 | grep cells; done; } | tee gpurun_out/r4_wide_small.txt
