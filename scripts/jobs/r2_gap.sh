/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/ubench/launch_gap.hip -o /tmp/lg && /tmp/lg
