timeout 900 python -m pytest tests/test_pipeline_dropin.py tests/test_meta_clustering.py -m gpu -x -q 2>&1 | tail -2
timeout 600 python scripts/pipeline_bench.py --fovs 6 2>&1 | tail -1
timeout 900 python scripts/pipeline_bench.py --fovs 40 2>&1 | tail -1
