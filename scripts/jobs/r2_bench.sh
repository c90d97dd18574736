python bench.py --steps 5 --warmup 2 > gpurun_out/bench_cfg2.json 2> gpurun_out/bench_cfg2.err; tail -c 3000 gpurun_out/bench_cfg2.json; tail -3 gpurun_out/bench_cfg2.err
for c in cfg4 cfg5 cfg3; do python bench.py --config $c --steps 3 --warmup 1 --no-pmc 2>&1 | tail -1 | cut -c1-1500; done
python bench.py --gpus 2 --steps 1; echo "rc=$?"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
