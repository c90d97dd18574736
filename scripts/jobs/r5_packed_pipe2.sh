# Round 5: packed-K filter pipelined inside the wave, second pass: a block's fragments kept in registers (PXSOM_PACKED_PIPE=2) and 512-thread workgroups
R=$GRAFT_REPO_ROOT
cd $R; O=gpurun_out/r5_packed_pipe; mkdir -p $O
probe() {
  PXSOM_HIPCC_EXTRA="$2" python -c "from ark_analysis_amd import _build; _build.build(force=True)" > $O/build_$1.log 2>&1 || { tail -5 $O/build_$1.log; return; }
  for bd in 1024 512; do
  (cd /tmp && export TMPDIR=/tmp && PXSOM_PACKED_BD=$bd PXSOM_HIPCC_EXTRA="$2" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk_$1_$bd -o pk -- python $R/scripts/debug/packed_filter_probe.py > /dev/null 2>&1)
  python scripts/prof_summarize.py /tmp/pk_$1_$bd $O/packed_probe_trace_$1_$bd.txt > /dev/null
  echo "== $1 threads $bd"; grep -i "packed" $O/packed_probe_trace_$1_$bd.txt | cut -c1-70,96-170 | tail -1
  done
}
{
probe pipe2 "-DPXSOM_PACKED_PIPE=2"
probe pipe1 "-DPXSOM_PACKED_PIPE=1"
probe plain "-DPXSOM_PACKED_PIPE=0"
} 2>&1 | tee $O/summary2.txt
