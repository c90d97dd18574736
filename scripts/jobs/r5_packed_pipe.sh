# Round 5: packed-K filter, MFMAs of one tile pair under the top-2 of the other inside every wave (PXSOM_PACKED_PIPE): parity, then same-box
# A/B on the config 5 probe (4.2 M x 40 binary16 rows, 400 nodes) against the plain loop, and the config 5 line
R=$GRAFT_REPO_ROOT
cd $R; O=gpurun_out/r5_packed_pipe; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_som_kernels.py tests/test_gpu_wide_rows.py -m gpu -x -q -k "packed or config5 or binary16 or f16 or half" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
PXSOM_FUZZ_DTYPE=f16 PXSOM_FUZZ_CASES=400 timeout 900 python -m pytest tests/test_gpu_fuzz_parity.py -m gpu -x -q -k "fuzz_assign" 2>&1 | tail -2
probe() {
  if [ -n "$2" ]; then PXSOM_HIPCC_EXTRA="$2" python -c "from ark_analysis_amd import _build; _build.build(force=True)" > $O/build_$1.log 2>&1 || { tail -5 $O/build_$1.log; return; }; fi
  (cd /tmp && export TMPDIR=/tmp && PXSOM_HIPCC_EXTRA="$2" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk_$1 -o pk -- python $R/scripts/debug/packed_filter_probe.py > /dev/null 2>&1)
  python scripts/prof_summarize.py /tmp/pk_$1 $O/packed_probe_trace_$1.txt > /dev/null
  echo "== $1"; grep -i "packed" $O/packed_probe_trace_$1.txt | cut -c1-70,96-170 | tail -1
}
line() { PXSOM_HIPCC_EXTRA="$3" python bench.py --config $2 --steps 3 --warmup 1 --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$1 $2', d['value'], d['ms_per_step'], d['phases_ms'])"; }
{
probe pipe_valu4 ""
line pipe_valu4 cfg5 ""
probe pipe_valu3 "-DPXSOM_PACKED_PIPE_VALU=3"
probe pipe_valu6 "-DPXSOM_PACKED_PIPE_VALU=6"
probe plain "-DPXSOM_PACKED_PIPE=0"
line plain cfg5 "-DPXSOM_PACKED_PIPE=0"
} 2>&1 | tee $O/summary.txt
