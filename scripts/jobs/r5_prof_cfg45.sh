# kernel traces of the config 4 / config 5 bench steps (generic training route) -> gpurun_out/r5_cfg45/
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r5_cfg45
cd /tmp && export TMPDIR=/tmp
for c in cfg4 cfg5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$c -o t -- python $R/bench.py --config $c --steps 3 --warmup 1 --no-pmc --no-cpu-baseline > /dev/null 2>&1
  python $R/scripts/prof_summarize.py /tmp/p_$c $R/gpurun_out/r5_cfg45/kernel_trace_$c.txt bmu_ cluster_sums batch_ centring stats_ fill memset > /dev/null
  echo "== $c"; head -30 $R/gpurun_out/r5_cfg45/kernel_trace_$c.txt | cut -c1-70,100-190
done
