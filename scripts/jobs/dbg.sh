for i in 1 2; do
PXSOM_SUMS_SORTED=1 python scripts/debug/k8_cost.py 2>&1 | tail -2 | head -1
PXSOM_SUMS_SORTED=0 python scripts/debug/k8_cost.py 2>&1 | tail -2 | head -1
done
