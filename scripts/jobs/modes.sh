for d in 0 2 3; do for m in 0 2; do echo "DMA=$d"; PXSOM_FILTER_DMA=$d TRAINED=1 PXSOM_FILTER_MODE=$m python scripts/assign_microbench.py 2>/dev/null | tail -1; done; done
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-online | grep -o '"phases_ms".*"frac": [0-9.]*'
