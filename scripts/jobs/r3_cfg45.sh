# kernel traces of configs 4 and 5 (per (kernel, grid) durations) -> gpurun_out/r3_cfg45/
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3_cfg45
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in cfg4 cfg5; do
  RAW=/tmp/r3_$cfg; mkdir -p $RAW
  rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o trace -- python $R/bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-online --no-pmc > $OUT/bench_$cfg.log 2>&1
  python $R/scripts/prof_summarize.py $RAW/trace $OUT/kernel_trace_$cfg.txt > /dev/null
  tail -1 $OUT/bench_$cfg.log | cut -c1-400
  grep -A12 "kernel stats" $OUT/kernel_trace_$cfg.txt | cut -c1-70,100-175
done
