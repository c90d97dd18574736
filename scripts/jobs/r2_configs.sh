for c in cfg2 cfg3 cfg4 cfg5; do
  timeout 900 python bench.py --config $c --steps 5 --warmup 2 --no-pmc > gpurun_out/bench_$c.json 2> gpurun_out/bench_$c.err
  python -c "
import json; d=json.load(open('gpurun_out/bench_$c.json')); print('$c', d['value'], d['unit'], d['ms_per_step'], d['phases_ms'], d['roofline']['frac'], d['roofline_step']['frac'], d.get('batch_train',{}).get('max_rel_err'), d.get('cpu_baseline',{}).get('value'))"
done
