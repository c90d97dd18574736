# sweep after the last kernel changes of the round: default split of the exact kernels, then every list through the
# screened kernel, then binary16 only (fixed-point mean table)
for seed in 31337 4242; do
  PXSOM_FUZZ_SEED=$seed PXSOM_FUZZ_CASES=1500 timeout 2400 python -m pytest tests/test_gpu_fuzz_parity.py -x -q 2>&1 | tail -2
done
PXSOM_SCREEN_MIN_ROWS=1 PXSOM_FUZZ_SEED=777 PXSOM_FUZZ_CASES=1500 timeout 2400 python -m pytest tests/test_gpu_fuzz_parity.py -x -q 2>&1 | tail -2
PXSOM_FUZZ_DTYPE=f16 PXSOM_FUZZ_SEED=555 PXSOM_FUZZ_CASES=1500 timeout 2400 python -m pytest tests/test_gpu_fuzz_parity.py -x -q 2>&1 | tail -2
