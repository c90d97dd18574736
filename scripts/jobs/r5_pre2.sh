# Round 5, item 7: create_pixel_matrix with the tables of several FOVs assembled side by side, page-locked row blocks, eight writers
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5_pre2
timeout 1200 python -m pytest tests/test_pipeline_dropin.py tests/test_gpu_preprocessing.py -m gpu -x -q > gpurun_out/r5_pre2/pytest.log 2>&1; tail -3 gpurun_out/r5_pre2/pytest.log
for n in 10 30; do python scripts/debug/create_pixel_matrix_timeline.py --fovs $n 2>&1 | tail -3; done | tee gpurun_out/r5_pre2/timeline.txt
python scripts/preprocess_bench.py 2>/dev/null | tail -2 | tee gpurun_out/r5_pre2/preprocess_bench.txt
