timeout 900 python -m pytest tests/test_pipeline_dropin.py -m gpu -x -q 2>&1 | tail -5
python scripts/pipeline_bench.py --fovs 4 --scratch /dev/shm 2>&1 | tail -1
python scripts/pipeline_bench.py --fovs 8 --scratch /tmp 2>&1 | tail -1
