cd /tmp && export TMPDIR=/tmp
for c in cfg5 cfg4; do
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$c -o t -- python $GRAFT_REPO_ROOT/bench.py --config $c --fovs-per-gpu 4 --steps 3 --warmup 1 --no-pmc > /dev/null 2>&1
python $GRAFT_REPO_ROOT/scripts/prof_summarize.py /tmp/p_$c /tmp/p_$c/sum.txt > /dev/null; echo "== $c"; grep -A14 "per (kernel, grid)" /tmp/p_$c/sum.txt | cut -c1-60,88-170
done
