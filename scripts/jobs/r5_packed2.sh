# Round 5, item 3: packed-K filter in two stages (Wh MFMAs alone first, deferred full search): parity + same-box A/B on the config 5 shape
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_packed2; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_som_kernels.py tests/test_gpu_wide_rows.py -m gpu -x -q -k "packed or config5 or binary16 or f16 or half" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
PXSOM_FUZZ_DTYPE=f16 PXSOM_FUZZ_CASES=400 timeout 900 python -m pytest tests/test_gpu_fuzz_parity.py -m gpu -x -q -k "fuzz_assign" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  PXSOM_PACKED_TWO=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk$v -o pk -- python $GRAFT_REPO_ROOT/scripts/debug/packed_filter_probe.py > /dev/null 2>&1
  python $GRAFT_REPO_ROOT/scripts/prof_summarize.py /tmp/pk$v $GRAFT_REPO_ROOT/$O/packed_probe_trace_two$v.txt > /dev/null
  echo "== PXSOM_PACKED_TWO=$v"; grep -i "packed\|exact" $GRAFT_REPO_ROOT/$O/packed_probe_trace_two$v.txt | cut -c1-70,96-170 | tail -3
done
