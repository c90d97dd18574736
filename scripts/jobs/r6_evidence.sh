# Round 6: the two evidence holes of round 5's review -- a cfg1 line (the plumbing config) and the end-to-end pipeline functions at HEAD
R=$GRAFT_REPO_ROOT
cd $R; O=gpurun_out/r6_evidence; mkdir -p $O
python bench.py --config cfg1 --steps 5 --warmup 1 > $O/bench_cfg1.json 2> $O/bench_cfg1.err; tail -1 $O/bench_cfg1.json | cut -c1-1500
python scripts/pipeline_bench.py --fovs 40 > $O/cluster_pixels.json 2> $O/cluster_pixels.err; tail -1 $O/cluster_pixels.json
python scripts/preprocess_bench.py --fovs 30 > $O/create_pixel_matrix.json 2> $O/create_pixel_matrix.err; tail -1 $O/create_pixel_matrix.json
nproc
