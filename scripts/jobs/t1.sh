timeout 600 python -m pytest tests/test_gpu_preprocessing.py tests/test_pipeline_dropin.py -m gpu -x -q 2>&1 | grep -v "^$" | tail -25
