# Round 5, item 6: binary64 rows through the two-tile one-pass kernel (no spills) against bmu_filter_fast<double> (28 - 138 spilled VGPRs)
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r5_f64; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_som_kernels.py tests/test_gpu_fuzz_parity.py tests/test_pipeline_dropin.py -m gpu -x -q -k "sums or one_pass or deferred or means or fuzz_assign or vouched or pipeline" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
PXSOM_FUZZ_DTYPE=f64 PXSOM_FUZZ_CASES=200 timeout 900 python -m pytest tests/test_gpu_fuzz_parity.py -m gpu -x -q -k "fuzz_assign" > $O/fuzz_f64.log 2>&1; tail -2 $O/fuzz_f64.log
for v in 1 0 1 0; do echo "== PXSOM_ONEPASS_F64=$v"; PXSOM_ONEPASS_F64=$v python scripts/debug/f64_assign_probe.py 2>/dev/null; done | tee $O/f64_assign_probe.txt
