# packed-K filter (config 5 shape) with 512 / 1024 threads per workgroup: kernel durations + MFMA / VALU counters -> gpurun_out/r4_packed_bd.txt
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r4_packed_bd.txt
mkdir -p $R/gpurun_out; : > $OUT
cd /tmp && export TMPDIR=/tmp
for bd in 512 1024; do
  rm -rf /tmp/pk
  PXSOM_PACKED_BD=$bd rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o t -- python $R/scripts/debug/packed_filter_probe.py > /dev/null 2>&1
  python $R/scripts/prof_summarize.py /tmp/pk /tmp/pk/sum.txt bmu_filter_packed bmu_exact > /dev/null
  echo "=== threads per workgroup $bd" >> $OUT
  grep -E "bmu_filter_packed|bmu_exact" /tmp/pk/sum.txt | tail -3 | cut -c1-60,100-190 >> $OUT
done
cat $OUT
