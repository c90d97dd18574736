PXSOM_FUZZ_CASES=${1:-150} timeout 2400 python -m pytest tests/test_gpu_fuzz_parity.py -x -q 2>&1 | tail -30
