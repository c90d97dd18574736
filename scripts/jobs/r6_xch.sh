# Round 6: exchange tests (mixed routes, late peer), distributed tests, 2- and 8-rank dry bench lines
R=$GRAFT_REPO_ROOT
cd $R; O=gpurun_out/r6_xch; mkdir -p $O
timeout 2700 python -m pytest tests/test_gpu_exchange.py tests/test_gpu_distributed.py tests/test_gpu_bench_multirank.py -m gpu -x -q > $O/pytest.log 2>&1; tail -15 $O/pytest.log
