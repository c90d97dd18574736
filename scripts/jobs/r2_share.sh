# update + prepare kernel on several workgroups (output shared out by node blocks): parity, then configs 4 and 5 with / without
timeout 900 python -m pytest tests/test_gpu_som_kernels.py tests/test_gpu_fuzz_parity.py tests/test_gpu_distributed.py tests/test_gpu_exchange.py -m gpu -x -q 2>&1 | tail -2 | head -1
PXSOM_FUZZ_CASES=400 timeout 1200 python -m pytest tests/test_gpu_fuzz_parity.py -m gpu -x -q -k "batch" 2>&1 | tail -1
for share in 1 0; do for cfg in cfg4 cfg5; do PXSOM_UPDATE_SHARE=$share python bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('share=$share $cfg', d['value'], d['ms_per_step'], d['phases_ms'])"; done; done
