bash scripts/jobs/ubench6.sh | grep "update+prep"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-online | grep -o '"value": [0-9.]*\|"phases_ms".*"frac": [0-9.]*'
