# Round profile: rocprofv3 kernel trace + stats of the default bench command, and separate PMC passes
# (FETCH_SIZE / WRITE_SIZE / SQ) -- summaries only are copied back (gpurun_out/prof_round/).
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_round
RAW=/tmp/prof_round_raw
mkdir -p $OUT $RAW
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-online --no-pmc --no-operating-range"
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o trace -- $CMD > $OUT/bench_under_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $RAW/pmc_fetch -o pmc -- $CMD > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $RAW/pmc_write -o pmc -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $RAW/pmc_sq -o pmc -- $CMD > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $RAW/pmc_sq2 -o pmc -- $CMD > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $RAW/pmc_tcc -o pmc -- $CMD > /dev/null 2>&1
python $R/scripts/prof_summarize.py $RAW/trace $OUT/kernel_trace_stats.txt > /dev/null
for p in pmc_fetch pmc_write pmc_sq pmc_sq2 pmc_tcc; do python $R/scripts/prof_summarize.py $RAW/$p $OUT/${p}.txt > /dev/null; done
grep -h "bmu_filter" $OUT/pmc_fetch.txt $OUT/pmc_write.txt | grep -E "SIZE" | cut -c1-40,88-170
tail -1 $OUT/bench_under_trace.log | cut -c1-300
