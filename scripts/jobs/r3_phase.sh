C=ark_analysis_amd/csrc
mkdir -p gpurun_out/r3_phase
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -Iinclude -I$C scripts/ubench/step_phase_timing.hip $C/pxsom_api.hip $C/pxsom_assign_filter.hip $C/pxsom_assign_filter_acc.hip -ffinite-math-only -o /tmp/spt 2>&1 | grep -E "error"
/tmp/spt 1 144 | tee gpurun_out/r3_phase/tail_steps.txt
/tmp/spt 1 64 | tee gpurun_out/r3_phase/steps64.txt
/tmp/spt 0 6 | tee gpurun_out/r3_phase/head_steps.txt
/tmp/spt 1 6 | tee gpurun_out/r3_phase/head_steps_tpw1.txt
