set -x
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof1
RAW=/tmp/prof_raw
mkdir -p $OUT $RAW
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-online"
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o trace -- $CMD > $OUT/trace_run.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $RAW/pmc_fetch -o pmc -- $CMD > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $RAW/pmc_write -o pmc -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $RAW/pmc_sq -o pmc -- $CMD > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d $RAW/pmc_sq2 -o pmc -- $CMD > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $RAW/pmc_tcc -o pmc -- $CMD > /dev/null 2>&1
find $RAW -type f | head -40
python $R/scripts/prof_summarize.py $RAW/trace $OUT/trace_summary.txt
for p in pmc_fetch pmc_write pmc_sq pmc_sq2 pmc_tcc; do python $R/scripts/prof_summarize.py $RAW/$p $OUT/${p}_summary.txt > /dev/null; done
cat $OUT/pmc_*_summary.txt | grep -E "counters per|bmu_filter" | head -60
rocprofv3 -L 2>/dev/null | grep -E "^\s*(gpu-agent|Name)|FETCH_SIZE|WRITE_SIZE|MfmaUtil|VALUBusy" | head -20
