timeout 600 python -m pytest tests/test_pipeline_dropin.py tests/test_gpu_preprocessing.py -m gpu -x -q 2>&1 | tail -2
python scripts/preprocess_bench.py --fovs 6 --scratch /dev/shm 2>&1 | tail -1
