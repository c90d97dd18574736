# one-pass kernel: does a deeper row prefetch pay?  builds: product / third row set (4 dwords spilled) / no bias registers (timing only,
# wrong labels) / both -> gpurun_out/r4_acc_depth.txt
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r4_acc_depth.txt
mkdir -p $R/gpurun_out; : > $OUT
cd $R
for flags in "" "-DPXSOM_ACC_ABL_DEPTH2" "-DPXSOM_ACC_ABL_NOBIAS" "-DPXSOM_ACC_ABL_NOBIAS -DPXSOM_ACC_ABL_DEPTH2"; do
  export PXSOM_HIPCC_EXTRA="$flags"
  python -c "
from ark_analysis_amd import _build
_build.build()" > /dev/null 2>&1
  echo "=== flags: '$flags'" >> $OUT
  for rep in 1 2; do
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-online --no-pmc --no-operating-range 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['phases_ms'])" >> $OUT
  done
done
cat $OUT
