# one-pass kernel with three workgroups per CU (168 VGPRs, no transposed codebook copy in LDS): parity + same-box A/B -> gpurun_out/r4_wgs3.txt
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
for flags in "" "-DPXSOM_FAST_WGS=3" "" "-DPXSOM_FAST_WGS=3"; do
  export PXSOM_HIPCC_EXTRA="$flags"
  python -c "
from ark_analysis_amd import _build
_build.build()" > /dev/null 2>&1
  echo "=== flags '$flags'"
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-online --no-pmc --no-operating-range 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['phases_ms'])"
done | tee gpurun_out/r4_wgs3.txt
timeout 1200 python -m pytest tests/test_gpu_som_kernels.py tests/test_gpu_fuzz_parity.py -m gpu -q 2>&1 | tail -2 | tee -a gpurun_out/r4_wgs3.txt
