C=ark_analysis_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -Iinclude -I$C scripts/ubench/assign_phase_timing.hip $C/pxsom_api.hip $C/pxsom_assign_filter.hip $C/pxsom_assign_filter_acc.hip -ffinite-math-only -o /tmp/apt 2>&1 | grep -E "error" ; /tmp/apt | tail -9; /tmp/apt 40 400 | tail -9
