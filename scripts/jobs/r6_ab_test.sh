# A/B of variant builds + parity tests of the default build
R=$GRAFT_REPO_ROOT
cd $R; O=gpurun_out/r6_ab; mkdir -p $O
timeout 1700 python -m pytest tests/test_gpu_som_kernels.py tests/test_gpu_fuzz_parity.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
bash scripts/jobs/r6_ab.sh
