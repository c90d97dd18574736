# LDS counters of the fused step and the one-pass kernel in the default bench -> gpurun_out/r4_lds_pmc.txt
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-online --no-pmc --no-operating-range"
rm -rf /tmp/pmc_a /tmp/pmc_b
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_a -o pmc -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS --kernel-trace --output-format csv -d /tmp/pmc_b -o pmc -- $CMD > /dev/null 2>&1
mkdir -p $R/gpurun_out
python $R/scripts/prof_summarize.py /tmp/pmc_a $R/gpurun_out/r4_lds_pmc_a.txt > /dev/null
python $R/scripts/prof_summarize.py /tmp/pmc_b $R/gpurun_out/r4_lds_pmc_b.txt > /dev/null
grep -E "batch_step_kernel|bmu_filter_fast" $R/gpurun_out/r4_lds_pmc_a.txt | cut -c1-44,96-260 | head -40
grep -E "batch_step_kernel|bmu_filter_fast" $R/gpurun_out/r4_lds_pmc_b.txt | cut -c1-44,96-260 | head -40
