# Whole GPU suite at HEAD (binary64 routing, lean gain loop), then the training pass per kernel: round 4's tree against HEAD, same box
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_train_ab2; mkdir -p $O; cd $R
timeout 3000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
PXSOM_FUZZ_CASES=200 timeout 1500 python -m pytest tests/test_gpu_fuzz_parity.py -m gpu -x -q > $O/fuzz200.log 2>&1; tail -2 $O/fuzz200.log
cd /tmp && export TMPDIR=/tmp
for tree in _r4ref . _r4ref .; do
  name=$( [ "$tree" = "." ] && echo head || echo round4 )
  cd $R/$tree; python -c "from ark_analysis_amd import _build; _build.build()" > /dev/null 2>&1
  cd /tmp; rm -rf /tmp/tr_$name; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$name -o t -- python $R/$tree/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-online --no-pmc --no-operating-range > $O/bench_$name.log 2>&1
  python $R/scripts/prof_summarize.py /tmp/tr_$name $O/trace_$name.txt > /dev/null
  echo "== $name"; grep -E "batch_step_kernel|batch_update_kernel|centring|bmu_filter_fast" $O/trace_$name.txt | head -8 | cut -c1-62,96-150
done 2>&1 | tee $O/summary.txt
