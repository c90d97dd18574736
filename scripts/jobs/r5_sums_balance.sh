# Round 5: atomic per-cluster sums kernel, wide tables: workgroup count from the balance of flush and LDS atomics (PXSOM_SUMS_BALANCE=0: the full grid)
R=$GRAFT_REPO_ROOT
cd $R; O=gpurun_out/r5_sums_balance; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_som_kernels.py tests/test_gpu_wide_rows.py tests/test_gpu_schedule.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
line() { PXSOM_SUMS_BALANCE=$1 python bench.py --config $2 --steps 3 --warmup 1 --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('balance=$1 $2', d['value'], d['ms_per_step'], d['phases_ms'])"; }
{
for c in cfg4 cfg5; do line 0 $c; line 1 $c; line 0 $c; line 1 $c; done
cd /tmp && export TMPDIR=/tmp
for b in 0 1; do
  PXSOM_SUMS_BALANCE=$b rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sb$b -o t -- python $R/bench.py --config cfg4 --steps 3 --warmup 1 --no-pmc --no-cpu-baseline > /dev/null 2>&1
  python $R/scripts/prof_summarize.py /tmp/sb$b $R/$O/kernel_trace_cfg4_balance$b.txt cluster_sums > /dev/null
  echo "== cfg4 balance=$b"; grep cluster_sums $R/$O/kernel_trace_cfg4_balance$b.txt | cut -c1-60,96-190 | tail -6
done
} 2>&1 | tee $O/summary.txt
