python scripts/shape_bench.py 2>/dev/null | tail -4
