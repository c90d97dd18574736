# stage 2 of the one-pass kernel: node blocks whose MFMA chains run side by side (PXSOM_FINE_BLOCKS = 2 / 4 / 7) -> gpurun_out/r4_fine_blocks.txt
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
for fb in 2 4 7; do
  export PXSOM_HIPCC_EXTRA="-DPXSOM_FINE_BLOCKS=$fb"
  python -c "
from ark_analysis_amd import _build
_build.build()" > /dev/null 2>&1
  echo "=== PXSOM_FINE_BLOCKS=$fb"
  for rep in 1 2; do python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-online --no-pmc --no-operating-range 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['phases_ms'])"; done
done | tee gpurun_out/r4_fine_blocks.txt
