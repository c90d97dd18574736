R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r3_pipe
python scripts/pipeline_bench.py --fovs 40 > gpurun_out/r3_pipe/cluster_pixels.json 2> gpurun_out/r3_pipe/cluster_pixels.err; tail -1 gpurun_out/r3_pipe/cluster_pixels.json
python scripts/preprocess_bench.py --fovs 10 > gpurun_out/r3_pipe/create_pixel_matrix.json 2> gpurun_out/r3_pipe/create_pixel_matrix.err; tail -1 gpurun_out/r3_pipe/create_pixel_matrix.json
