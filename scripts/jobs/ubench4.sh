/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/ubench/f64_single_wave.hip -o /tmp/f64sw && /tmp/f64sw
