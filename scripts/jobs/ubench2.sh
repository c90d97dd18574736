cd scripts/ubench && hipcc --offload-arch=gfx950 -O3 valu_cost.hip -o /tmp/valu_cost 2>&1 | grep -E "error" ; /tmp/valu_cost
