cd scripts/ubench && hipcc --offload-arch=gfx950 -O3 overlap.hip -o /tmp/overlap 2>&1 | grep -E "error"; /tmp/overlap
