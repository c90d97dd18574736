# accumulating filter variants: parity subset + bench
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r3_acc2; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_som_kernels.py tests/test_gpu_fuzz_parity.py -x -q -k "one_pass or assign_sums or assign_means or batch or fuzz" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for i in 1 2 3; do python bench.py --steps 10 --warmup 2 --no-pmc --no-cpu-baseline --no-online --no-operating-range 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['phases_ms'])" | tee -a $O/bench.txt; done
