PXSOM_FUZZ_DTYPE=f64 PXSOM_FUZZ_SEED=9001 PXSOM_FUZZ_CASES=1500 timeout 2400 python -m pytest tests/test_gpu_fuzz_parity.py -x -q 2>&1 | tail -1
PXSOM_FUZZ_DTYPE=f32 PXSOM_FUZZ_SEED=9002 PXSOM_FUZZ_CASES=1500 timeout 2400 python -m pytest tests/test_gpu_fuzz_parity.py -x -q 2>&1 | tail -1
PXSOM_SCREEN_MIN_ROWS=64 PXSOM_FUZZ_SEED=9003 PXSOM_FUZZ_CASES=1500 timeout 2400 python -m pytest tests/test_gpu_fuzz_parity.py -x -q 2>&1 | tail -1
