# Round 5: one-pass kernel (two waves per SIMD) with the four labels of a trip exchanged by v_permlane swaps instead of ds_bpermute
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r5_swaplabels; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_som_kernels.py tests/test_gpu_fuzz_parity.py -m gpu -x -q -k "sums or one_pass or deferred or means or fuzz_assign or vouched" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for v in 0 1 0; do
  PXSOM_ONEPASS=$v python bench.py --no-cpu-baseline --no-online --no-pmc --no-operating-range 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('ONEPASS=$v',d['value'],d['ms_per_step'],d['phases_ms'],d['roofline']['frac'])"
done
