# Round 5, item 7: where create_pixel_matrix's time goes (scripts/debug/create_pixel_matrix_timeline.py), 10 and 30 FOVs
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5_pre_tl
python -c "import os; print('host cores', os.cpu_count())"; df -h /tmp | tail -1
for n in 10 30; do python scripts/debug/create_pixel_matrix_timeline.py --fovs $n 2>&1 | tail -3; done | tee gpurun_out/r5_pre_tl/timeline.txt
