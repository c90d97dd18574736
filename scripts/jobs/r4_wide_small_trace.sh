cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -o t -- python $GRAFT_REPO_ROOT/scripts/debug/small_cell_table_probe.py > /dev/null 2>&1
python $GRAFT_REPO_ROOT/scripts/prof_summarize.py /tmp/ps /tmp/ps/sum.txt batch_step_wide centring batch_update bmu_ cluster_sums > /dev/null
grep -E "batch_step_wide|centring|batch_update|bmu_|cluster_sums" /tmp/ps/sum.txt | cut -c1-60,100-190 | head -40
