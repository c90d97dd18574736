# packed-K filter with 1024 threads per workgroup: parity (kernel tests + sweep), config 5 with 512 / 1024 -> gpurun_out/r4_packed_cfg5.txt
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_som_kernels.py tests/test_gpu_fuzz_parity.py -m gpu -q 2>&1 | tail -3 | tee gpurun_out/r4_packed_cfg5.txt
for bd in 512 1024; do
  PXSOM_PACKED_BD=$bd python bench.py --config cfg5 --steps 3 --warmup 1 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('cfg5 threads $bd',d['value'],d['ms_per_step'],d['phases_ms'])" | tee -a gpurun_out/r4_packed_cfg5.txt
done
python bench.py --config cfg4 --steps 3 --warmup 1 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('cfg4',d['value'],d['ms_per_step'],d['phases_ms'])" | tee -a gpurun_out/r4_packed_cfg5.txt
