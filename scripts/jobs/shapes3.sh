timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -1
python scripts/shape_bench.py 2>/dev/null | tail -4 | cut -c1-120
