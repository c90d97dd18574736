R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r3_tail
for t in 1 2 4; do
PXSOM_STEP_TPW_SMALL=$t python bench.py --steps 5 --warmup 2 --no-pmc --no-cpu-baseline --no-online --no-operating-range 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('small tpw $t',d['value'],d['ms_per_step'],d['phases_ms'])" | tee -a gpurun_out/r3_tail/bench.txt
done
