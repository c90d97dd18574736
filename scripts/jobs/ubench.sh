cd scripts/ubench && hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o /tmp/mfma_rate 2>/dev/null && /tmp/mfma_rate
