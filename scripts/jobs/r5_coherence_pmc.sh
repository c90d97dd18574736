# Round 5: LDS counters of the one-pass kernel on rows as generated / sorted by label, with the row-axis sums (HEAD) and without (-DPXSOM_ADD_SCAN=0)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_coherence; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
pass() {   # name, order, extra flags
  rm -rf /tmp/pmc_$1
  PXSOM_HIPCC_EXTRA="$3" rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_$1 -o pmc -- python $R/scripts/debug/label_coherence_pmc_target.py $2 > /dev/null 2>&1
  python $R/scripts/prof_summarize.py /tmp/pmc_$1 $O/pmc_lds_$1.txt > /dev/null
  echo "== $1"; grep -E "bmu_filter_fast" $O/pmc_lds_$1.txt | cut -c1-44,96-260 | head -12
}
{
pass scan_generated generated ""
pass scan_sorted sorted ""
(cd $R && PXSOM_HIPCC_EXTRA="-DPXSOM_ADD_SCAN=0" python -c "from ark_analysis_amd import _build; _build.build(force=True)" > $O/build_off.log 2>&1)
pass plain_generated generated "-DPXSOM_ADD_SCAN=0"
pass plain_sorted sorted "-DPXSOM_ADD_SCAN=0"
} 2>&1 | tee $O/pmc_lds_summary.txt
