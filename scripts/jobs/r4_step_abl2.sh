# fused step: does the windowed-update code cost the BMU-only steps anything by being there?  (timing build: windowed steps wrong)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r4_step_abl2
for flags in "" "-DPXSOM_STEP_ABL_BMUONLY"; do
  export PXSOM_HIPCC_EXTRA="$flags"
  python -c "
from ark_analysis_amd import _build
_build.build()" > /dev/null 2>&1
  cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/tr
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o trace -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-online --no-pmc --no-operating-range > /dev/null 2>&1
  python $R/scripts/prof_summarize.py /tmp/tr /tmp/tr/sum.txt batch_step > /dev/null
  echo "=== flags '$flags'"; grep -E "batch_step_kernel" /tmp/tr/sum.txt | tail -9 | cut -c1-50,100-180
  cd $R
done | tee gpurun_out/r4_step_abl2/out.txt
