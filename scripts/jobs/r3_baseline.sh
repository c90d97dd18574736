# Round 3 start: the round-2 tree on this round's box -- default bench line (reference point for the round's changes).
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r3_base
cd $R && python bench.py --steps 5 --warmup 2 --no-pmc > gpurun_out/r3_base/bench_default.json 2> gpurun_out/r3_base/bench_default.err
tail -c 1500 gpurun_out/r3_base/bench_default.json
