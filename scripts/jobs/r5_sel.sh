# Round 5: the node-block select of the streamed / packed-K filters as a VOP3 pair on an SGPR mask (block_if_moved) against the
# compiler's VOP2 v_cndmask on VCC (-DPXSOM_SEL_VCC=1): parity, then same-box A/B on the config 5 probe and the config 4 / 5 lines
R=$GRAFT_REPO_ROOT
cd $R; O=gpurun_out/r5_sel; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_som_kernels.py tests/test_gpu_wide_rows.py -m gpu -x -q -k "packed or config5 or binary16 or f16 or half or wide or stream" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
PXSOM_FUZZ_DTYPE=f16 PXSOM_FUZZ_CASES=300 timeout 900 python -m pytest tests/test_gpu_fuzz_parity.py -m gpu -x -q -k "fuzz_assign" 2>&1 | tail -2
probe() {
  (cd /tmp && export TMPDIR=/tmp && PXSOM_HIPCC_EXTRA="$2" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk_$1 -o pk -- python $R/scripts/debug/packed_filter_probe.py > /dev/null 2>&1)
  python scripts/prof_summarize.py /tmp/pk_$1 $O/packed_probe_trace_$1.txt > /dev/null
  echo "== $1"; grep -i "packed\|exact" $O/packed_probe_trace_$1.txt | cut -c1-70,96-170 | tail -3
}
line() { PXSOM_HIPCC_EXTRA="$3" python bench.py --config $2 --steps 3 --warmup 1 --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$1 $2', d['value'], d['ms_per_step'], d['phases_ms'])"; }
{
probe sgpr ""
line sgpr cfg5 ""; line sgpr cfg4 ""
PXSOM_HIPCC_EXTRA="-DPXSOM_SEL_VCC=1" python -c "from ark_analysis_amd import _build; _build.build(force=True)" > $O/build_vcc.log 2>&1 || tail -5 $O/build_vcc.log
probe vcc "-DPXSOM_SEL_VCC=1"
line vcc cfg5 "-DPXSOM_SEL_VCC=1"; line vcc cfg4 "-DPXSOM_SEL_VCC=1"
} 2>&1 | tee $O/summary.txt
