# SQ counters of the packed-K filter (config 5 shape) -- profiles/r03/packed_filter_counters.txt was made with two experimental
# builds of round 3 (row prefetch + double-buffered fragments; one max per tile instead of the top-2) on top of this recipe
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r3_packed_pmc; mkdir -p $O
CMD="python $R/scripts/debug/packed_filter_probe.py"
rm -rf /tmp/pp_a /tmp/pp_b
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d /tmp/pp_a -o pmc -- $CMD > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d /tmp/pp_b -o pmc -- $CMD > /dev/null 2>&1
python $R/scripts/prof_summarize.py /tmp/pp_a $O/sq.txt bmu_filter_packed > /dev/null
python $R/scripts/prof_summarize.py /tmp/pp_b $O/sq2.txt bmu_filter_packed > /dev/null
cut -c1-60,88-190 $O/sq.txt $O/sq2.txt
