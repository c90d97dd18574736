set -x
python -c "import __graft_entry__ as g; g.smoke()"
python bench.py --steps 5 --warmup 2
