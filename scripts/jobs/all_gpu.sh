timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 5 --warmup 2 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; python -c "
import json; d=json.load(open('gpurun_out/bench_default.json')); print(d['value'], d['ms_per_step'], d['phases_ms'], d['roofline']['frac'], d['roofline'].get('traffic'), d.get('mfma_util',{}).get('value'), d['roofline_step']['frac'], d.get('batch_train'))"
