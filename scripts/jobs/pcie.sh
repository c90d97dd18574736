python scripts/pcie_rate.py 2>/dev/null | tail -1
