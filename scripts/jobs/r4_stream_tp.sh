# streamed filter on config 4 (four channel chunks): tiles per fragment read 2 (default) / 1 -> gpurun_out/r4_stream_tp.txt
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
for tp in 2 1; do
  export PXSOM_STREAM_TP=$tp
  python -c "
from ark_analysis_amd import _build
_build.build()" > /dev/null 2>&1
  python bench.py --config cfg4 --steps 3 --warmup 1 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('TP $tp',d['value'],d['ms_per_step'],d['phases_ms'])"
done | tee gpurun_out/r4_stream_tp.txt
