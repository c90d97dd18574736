for i in 1 2 3; do timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -1; done
for i in 1 2 3; do python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-online | grep -o '"value": [0-9.]*\|"train_batch": [0-9.]*\|"assign_filter_kernel": [0-9.]*' | tr '\n' ' '; echo; done
