# labels-only register-resident filter with the deferred full search, two / three workgroups per CU: parity + labels-only times -> gpurun_out/r4_plain_defer.txt
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
for flags in "" "-DPXSOM_PLAIN_WGS=3" "" "-DPXSOM_PLAIN_WGS=3"; do
  export PXSOM_HIPCC_EXTRA="$flags"
  python -c "
from ark_analysis_amd import _build
_build.build()" > /dev/null 2>&1
  echo "=== flags '$flags'"
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-online --no-pmc 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['phases_ms']['assign_filter_kernel']);print({k:(v['assign_ms'],v['labels_and_mean_table_one_pass_ms']) for k,v in d['operating_range'].items() if isinstance(v,dict)})"
done | tee gpurun_out/r4_plain_defer.txt
unset PXSOM_HIPCC_EXTRA
python -c "
from ark_analysis_amd import _build
_build.build()" > /dev/null 2>&1
timeout 1500 python -m pytest tests/test_gpu_som_kernels.py tests/test_gpu_fuzz_parity.py tests/test_pipeline_dropin.py -m gpu -q 2>&1 | tail -2 | tee -a gpurun_out/r4_plain_defer.txt
