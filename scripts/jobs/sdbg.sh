python scripts/debug/sums_dbg.py
