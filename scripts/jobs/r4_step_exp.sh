# where does a fused step's time go in the PRODUCT run?  The default bench under a kernel trace with parts of the step kernel
# switched off (builds with -DPXSOM_STEP_EXPERIMENT=n: results are wrong, durations are what is read) -> gpurun_out/r4_step_exp.txt
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
rm -f gpurun_out/r4_step_exp.txt
for m in 0 1 2 3; do
  if [ $m = 0 ]; then unset PXSOM_STEP_EXPERIMENT; else export PXSOM_STEP_EXPERIMENT=$m; fi
  python -c "from ark_analysis_amd import _build; _build.build(force=True)" > /dev/null 2>&1
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tr$m && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr$m -o trace -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-online --no-pmc --no-operating-range > /dev/null 2>&1 )
  python scripts/prof_summarize.py /tmp/tr$m /tmp/kts$m.txt > /dev/null
  echo "=== experiment $m (0: product; 1: listed rows not settled; 2: no table adds; 3: search of one node block only)" >> gpurun_out/r4_step_exp.txt
  grep -E "batch_step_kernel" /tmp/kts$m.txt | grep -v "StepArg " | cut -c1-40,96-200 >> gpurun_out/r4_step_exp.txt
done
unset PXSOM_STEP_EXPERIMENT
python -c "from ark_analysis_amd import _build; _build.build(force=True)" > /dev/null 2>&1
cat gpurun_out/r4_step_exp.txt
