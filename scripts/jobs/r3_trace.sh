# kernel trace of the default bench command (per (kernel, grid) durations) -> gpurun_out/r3_trace/
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3_trace
RAW=/tmp/r3_trace_raw
mkdir -p $OUT $RAW
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-online --no-pmc $BENCH_ARGS"
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o trace -- $CMD > $OUT/bench_under_trace.log 2>&1
python $R/scripts/prof_summarize.py $RAW/trace $OUT/kernel_trace_stats.txt > /dev/null
grep -E "batch_step|bmu_filter|cluster_sums|bmu_exact|bmu_prep" $OUT/kernel_trace_stats.txt | cut -c1-60,88-200 | head -60
