timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for m in 0 2; do TRAINED=1 PXSOM_FILTER_MODE=$m python scripts/assign_microbench.py 2>/dev/null | tail -1; done
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-online | grep -o '"phases_ms".*"frac": [0-9.]*'
