# Round 5: the scan path in the two-tile kernel (binary64 rows): parity, coherence probe on binary64 rows, binary64 / binary32 probe at HEAD
R=$GRAFT_REPO_ROOT
cd $R; O=gpurun_out/r5_scan; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_som_kernels.py tests/test_gpu_fuzz_parity.py -m gpu -x -q > $O/pytest64.log 2>&1; tail -2 $O/pytest64.log
PXSOM_FUZZ_DTYPE=f64 PXSOM_FUZZ_CASES=400 timeout 900 python -m pytest tests/test_gpu_fuzz_parity.py -m gpu -x -q 2>&1 | tail -1
python scripts/debug/label_coherence_probe.py f64 2>&1 | grep -v amdgpu.ids | tee $O/probe_f64_scan.txt
python scripts/debug/f64_assign_probe.py 2>/dev/null | tee $O/f64_assign_probe.txt
PXSOM_ONEPASS=1 python scripts/debug/label_coherence_probe.py 2>&1 | grep -v amdgpu.ids | head -4 | tee $O/probe_f32_two_tile.txt
