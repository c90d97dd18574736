cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4_pipe
timeout 900 python -m pytest tests/test_pipeline_dropin.py tests/test_gpu_fuzz_parity.py -m gpu -q -k "pipeline or arrow or cluster_pixels or label or front" > gpurun_out/r4_pipe/pytest.log 2>&1; tail -2 gpurun_out/r4_pipe/pytest.log
python scripts/debug/cluster_pixels_timeline.py --fovs 40 2>&1 | tail -4 | tee gpurun_out/r4_pipe/cluster_pixels_timeline.txt
python scripts/pipeline_bench.py --fovs 40 > gpurun_out/r4_pipe/cluster_pixels.json 2> gpurun_out/r4_pipe/cluster_pixels.err; tail -1 gpurun_out/r4_pipe/cluster_pixels.json
