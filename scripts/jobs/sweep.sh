python scripts/debug/assign_sweep.py
