timeout 900 python -m pytest tests/test_gpu_som_kernels.py -x -q -k "batch_train_steps or config5" 2>&1 | tail -8
for c in cfg5 cfg4; do python bench.py --config $c --steps 3 --warmup 1 --no-pmc 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['name'], d['value'], d['ms_per_step'], d['phases_ms'])"; done
