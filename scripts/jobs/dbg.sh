python scripts/debug/amb_per_step.py 2>&1 | tail -1
