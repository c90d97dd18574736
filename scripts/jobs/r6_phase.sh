# Round 6: per-phase stamps of the fused step at HEAD (tail steps of ~8.7 K rows, head steps)
C=ark_analysis_amd/csrc
O=gpurun_out/r6_phase; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -Iinclude -I$C scripts/ubench/step_phase_timing.hip $C/pxsom_api.hip $C/pxsom_assign_filter.hip $C/pxsom_assign_filter_acc.hip -ffinite-math-only -o /tmp/spt 2>&1 | grep -E "error"
/tmp/spt 1 120 | tee $O/tail_steps.txt
/tmp/spt 0 6 | tee $O/head_steps.txt
