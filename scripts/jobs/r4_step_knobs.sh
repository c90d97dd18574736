# step-kernel launch knobs on the round-4 kernel: tiles per wave of the small steps, workgroups per CU -> gpurun_out/r4_step_knobs.txt
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r4_step_knobs.txt
mkdir -p $R/gpurun_out; : > $OUT
cd $R
run() {
  echo "=== $*" >> $OUT
  for rep in 1 2; do
    env "$@" python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-online --no-pmc --no-operating-range 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['phases_ms'])" >> $OUT
  done
}
run PXSOM_NOP=1
run PXSOM_STEP_TPW_SMALL=2
run PXSOM_STEP_TPW_SMALL=4
run PXSOM_STEP_WGS_PER_CU=2
cat $OUT
