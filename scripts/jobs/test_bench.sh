set -x
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-online
