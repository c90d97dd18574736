timeout 600 python -m pytest tests/test_pipeline_dropin.py tests/test_gpu_preprocessing.py -m gpu -x -q 2>&1 | grep -E "^E|FAILED|Error" | head -20
