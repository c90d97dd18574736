R=$GRAFT_REPO_ROOT
RAW=/tmp/prof_steps
mkdir -p $RAW
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $RAW/t -o t -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-online > /dev/null 2>&1
python - <<'PY'
import csv, glob
f=glob.glob('/tmp/prof_steps/t/**/*kernel_trace.csv', recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
acc=[(int(r['Start_Timestamp']), int(r['End_Timestamp'])-int(r['Start_Timestamp'])) for r in rows if 'bmu_filter_fast' in r['Kernel_Name'] and 'Lb1E' in r['Kernel_Name']]
acc.sort()
print("accumulating-filter durations per step (us):", [round(d/1000,1) for _,d in acc])
PY
