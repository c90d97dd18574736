cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5_pre_prof
python scripts/debug/create_pixel_matrix_cprofile.py 20 2>&1 | tail -70 | tee gpurun_out/r5_pre_prof/cprofile.txt
