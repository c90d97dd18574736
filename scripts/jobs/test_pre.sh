timeout 900 python -m pytest tests/test_gpu_preprocessing.py -m gpu -x -q 2>&1 | tail -25
