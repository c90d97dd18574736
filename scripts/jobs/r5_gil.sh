# Round 5, f2: does the interpreter's GIL hand-over interval (sys.setswitchinterval, default 5 ms) pace create_pixel_matrix's calling thread?
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_gil; mkdir -p $O
python scripts/debug/create_pixel_matrix_timeline.py --fovs 4 > /dev/null 2>&1   # warm the page cache / imports
for s in 0.005 0.001 0.0002 0.00005; do
  echo "== switch interval $s"; PXSOM_SWITCH=$s python scripts/debug/create_pixel_matrix_timeline.py --fovs 30 2>&1 | tail -2
done 2>&1 | tee $O/summary.txt
