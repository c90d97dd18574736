R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc2
RAW=/tmp/prof_raw2
mkdir -p $OUT $RAW
cd /tmp && export TMPDIR=/tmp
export TRAINED=1
CMD="python $R/scripts/assign_microbench.py"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $RAW/a -o pmc -- $CMD > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $RAW/b -o pmc -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_INSTS_VMEM_RD SQ_IFETCH SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $RAW/c -o pmc -- $CMD > /dev/null 2>&1
for p in a b c; do python $R/scripts/prof_summarize.py $RAW/$p $OUT/${p}.txt bmu_filter > /dev/null; done
cat $OUT/a.txt $OUT/b.txt $OUT/c.txt | grep -E "bmu_filter" | grep -E " 131072 |196608|65536" | sed 's/_ZN9pxsom_bmu12_GLOBAL__N_1//' | cut -c1-30,85-170
