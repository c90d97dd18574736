# end-to-end pipeline functions -> gpurun_out/r4_pipe/
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r4_pipe
timeout 900 python -m pytest tests/test_pipeline_dropin.py tests/test_gpu_fuzz_parity.py -m gpu -q -k "pipeline or arrow or cluster_pixels or label or fuzz" > gpurun_out/r4_pipe/pytest.log 2>&1; tail -3 gpurun_out/r4_pipe/pytest.log
python scripts/pipeline_bench.py --fovs 40 > gpurun_out/r4_pipe/cluster_pixels.json 2> gpurun_out/r4_pipe/cluster_pixels.err; tail -1 gpurun_out/r4_pipe/cluster_pixels.json
python scripts/preprocess_bench.py --fovs 10 > gpurun_out/r4_pipe/create_pixel_matrix.json 2> gpurun_out/r4_pipe/create_pixel_matrix.err; tail -1 gpurun_out/r4_pipe/create_pixel_matrix.json
python scripts/debug/create_pixel_matrix_timeline.py --fovs 10 > gpurun_out/r4_pipe/timeline.txt 2>&1; tail -3 gpurun_out/r4_pipe/timeline.txt
nproc
