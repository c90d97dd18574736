# Round 5, item 2: the one-pass labels + table kernel at three waves per SIMD (pxsom_assign_onepass.h).  Parity of everything
# that goes through pxsom_assign_sums / means, then the same bench line with the kernel on and off (same box).
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r5_onepass; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_som_kernels.py tests/test_gpu_fuzz_parity.py -m gpu -x -q -k "sums or one_pass or deferred or means or fuzz_assign or vouched" > $O/pytest.log 2>&1; tail -6 $O/pytest.log
for v in 1 0 1 0; do
  PXSOM_ONEPASS=$v python bench.py --no-cpu-baseline --no-online --no-pmc --no-operating-range > $O/bench_$v.json 2> $O/bench_$v.err
  python -c "
import json;d=json.loads(open('$O/bench_$v.json').read().strip().splitlines()[-1]);print('ONEPASS=$v',d['value'],d['ms_per_step'],d['phases_ms'],d['roofline']['frac'])"
done
