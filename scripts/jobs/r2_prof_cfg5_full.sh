cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_cfg5 -o t -- python $GRAFT_REPO_ROOT/bench.py --config cfg5 --steps 2 --warmup 1 --no-pmc --no-cpu-baseline --no-online > /dev/null 2>&1
python $GRAFT_REPO_ROOT/scripts/prof_summarize.py /tmp/p_cfg5 /tmp/p_cfg5/sum.txt > /dev/null; grep -A14 "per (kernel, grid)" /tmp/p_cfg5/sum.txt | cut -c1-60,88-170
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/prof_cfg5 && cp /tmp/p_cfg5/sum.txt $GRAFT_REPO_ROOT/gpurun_out/prof_cfg5/kernel_trace_stats_cfg5_full.txt
