# Round 5, item 6: binary64 rows, labels only, through the two-tile kernel; the whole GPU suite behind it
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r5_f64b; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
PXSOM_FUZZ_DTYPE=f64 PXSOM_FUZZ_CASES=300 timeout 900 python -m pytest tests/test_gpu_fuzz_parity.py -m gpu -x -q -k "fuzz_assign or fuzz_batch" > $O/fuzz_f64.log 2>&1; tail -2 $O/fuzz_f64.log
for v in 1 0; do echo "== PXSOM_ONEPASS_F64=$v"; PXSOM_ONEPASS_F64=$v python scripts/debug/f64_assign_probe.py 2>/dev/null; done | tee $O/f64_assign_probe.txt
