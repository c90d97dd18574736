for seed in ${SEEDS:-11 2027 99991}; do
  PXSOM_FUZZ_SEED=$seed PXSOM_FUZZ_CASES=${1:-1500} timeout 2400 python -m pytest tests/test_gpu_fuzz_parity.py -x -q 2>&1 | tail -4
done
