R=$GRAFT_REPO_ROOT
RAW=/tmp/prof_raw3
mkdir -p $RAW $R/gpurun_out/prof_train
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o trace -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-online > /dev/null 2>&1
python $R/scripts/prof_summarize.py $RAW/trace $R/gpurun_out/prof_train/trace_summary.txt | grep -A12 "per (kernel, grid)" | cut -c1-60,88-170
