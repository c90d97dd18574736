# kernel trace of the default bench command (per (kernel, grid) durations) + phase stamps at HEAD
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r6_trace
RAW=/tmp/r6_trace_raw
mkdir -p $OUT $RAW
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-online --no-pmc --no-operating-range $BENCH_ARGS"
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o trace -- $CMD > $OUT/bench_under_trace.log 2>&1
python $R/scripts/prof_summarize.py $RAW/trace $OUT/kernel_trace_stats.txt > /dev/null
grep -E "batch_step|bmu_filter|cluster_sums|bmu_exact|bmu_prep|centring" $OUT/kernel_trace_stats.txt | cut -c1-60,88-200 | head -60
cd $R
C=ark_analysis_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -Iinclude -I$C scripts/ubench/step_phase_timing.hip $C/pxsom_api.hip $C/pxsom_assign_filter.hip $C/pxsom_assign_filter_acc.hip -ffinite-math-only -o /tmp/spt 2>&1 | grep -E "error"
/tmp/spt 1 120 | tee $OUT/tail_steps.txt | grep -A1 "step 116"
/tmp/spt 0 6 | tee $OUT/head_steps.txt | grep -A1 "step  [24]"
