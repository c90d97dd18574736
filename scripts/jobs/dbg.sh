python scripts/debug/chain_steps.py 2>&1 | tail -20
