R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r3_center; mkdir -p $O
for v in 0 1; do
  if [ $v = 1 ]; then export PXSOM_STEP_NO_CENTRE=1; fi
  echo "no_centre=$v"
  for i in 1 2; do python bench.py --steps 10 --warmup 2 --no-pmc --no-cpu-baseline --no-online --no-operating-range 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['phases_ms'])"; done
  bash scripts/jobs/r3_trace.sh > /dev/null 2>&1
  grep "batch_step_kernel" gpurun_out/r3_trace/kernel_trace_stats.txt | tail -7 | cut -c1-60,95-180
done
