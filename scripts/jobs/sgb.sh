cp ark_analysis_amd/libpxsom.so /tmp/orig.so
for v in 0 3 8; do cp scripts/ubench/libs/libpxsom_sgb$v.so ark_analysis_amd/libpxsom.so; echo "SGB_VALU=$v"; python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-online 2>/dev/null | grep -o '"assign_filter_kernel": [0-9.]*'; done
cp /tmp/orig.so ark_analysis_amd/libpxsom.so; echo "SGB_VALU=5 (default)"; python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-online 2>/dev/null | grep -o '"assign_filter_kernel": [0-9.]*'
