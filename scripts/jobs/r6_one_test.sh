# one test file / selection on the GPU: TESTS="tests/test_x.py -k name"
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/r6_one_test
timeout 1500 python -m pytest $TESTS -m gpu -x -q 2>&1 | tail -25 | tee gpurun_out/r6_one_test/pytest.log
