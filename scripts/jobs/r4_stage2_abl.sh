# one-pass kernel: what stage 2 costs -- product / stage 2 compiled out (rows it would settle keep their stage-1 label: timing only) /
# stage 2 compiled in but never taken -> gpurun_out/r4_stage2_abl.txt
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
for flags in "" "-DPXSOM_ABL_NO_STAGE2" "-DPXSOM_ABL_STAGE2_DORMANT"; do
  export PXSOM_HIPCC_EXTRA="$flags"
  python -c "
from ark_analysis_amd import _build
_build.build()" > /dev/null 2>&1
  echo "=== flags '$flags'"
  for rep in 1 2; do python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-online --no-pmc --no-operating-range 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['phases_ms'])"; done
done | tee gpurun_out/r4_stage2_abl.txt
