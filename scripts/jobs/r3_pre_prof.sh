# Round 3: rocprofv3 evidence for the pre-processing kernels (one 1024^2 x 22 FOV): kernel trace + FETCH / WRITE passes.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3_pre
RAW=/tmp/r3_pre_raw
mkdir -p $OUT $RAW
cd /tmp && export TMPDIR=/tmp
CMD="python $R/scripts/preprocess_kernels.py --reps 5"
$CMD > $OUT/plain.json 2>$OUT/plain.err
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o trace -- $CMD > $OUT/under_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $RAW/pmc_fetch -o pmc -- $CMD > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $RAW/pmc_write -o pmc -- $CMD > /dev/null 2>&1
KEEP="blur rowfilter block_scan q_hist q_select q_next q_finish q_init"
python $R/scripts/prof_summarize.py $RAW/trace $OUT/kernel_trace_stats.txt $KEEP > /dev/null
for p in pmc_fetch pmc_write; do python $R/scripts/prof_summarize.py $RAW/$p $OUT/${p}.txt $KEEP > /dev/null; done
cat $OUT/plain.json
