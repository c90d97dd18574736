cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/psh -o t -- python $GRAFT_REPO_ROOT/scripts/shape_bench.py > /dev/null 2>&1
python $GRAFT_REPO_ROOT/scripts/prof_summarize.py /tmp/psh /tmp/psh_sum.txt > /dev/null; grep -A60 "per (kernel, grid)" /tmp/psh_sum.txt | grep -v "at::\|rocprim\|elementwise" | cut -c1-64,88-175 | head -40
