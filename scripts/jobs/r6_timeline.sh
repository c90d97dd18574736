# Round 6: kernels of one training pass + assign in launch order (BENCH_ARGS="--config cfg4 ..."; N = kernels printed)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r6_timeline
RAW=/tmp/r6_timeline_raw
mkdir -p $OUT $RAW
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $RAW -o t -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-online --no-pmc --no-operating-range $BENCH_ARGS > $OUT/bench.log 2>&1
python $R/scripts/dev/trace_timeline.py $RAW ${N:-140} | tee $OUT/timeline.txt | tail -${N:-140}
