for c in cfg2 cfg5; do
timeout 1500 python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-online > gpurun_out/bench_$c.json 2> gpurun_out/bench_$c.err
python -c "
import json; d=json.load(open('gpurun_out/bench_$c.json')); r=d['roofline']; print('$c', d['value'], r['frac'], r.get('traffic'), r.get('traffic_source'), d.get('mfma_util',{}).get('value'), r.get('rows_per_launch', r.get('pixels_per_launch')))"
done
