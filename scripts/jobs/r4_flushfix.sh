# word-granular stagger of the step kernel's flush: phase stamps, parity, default bench + trace -> gpurun_out/r4_flushfix/
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r4_flushfix
cd $R
C=ark_analysis_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -Iinclude -I$C scripts/ubench/step_phase_timing.hip $C/pxsom_api.hip $C/pxsom_assign_filter.hip $C/pxsom_assign_filter_acc.hip -ffinite-math-only -o /tmp/spt 2>&1 | grep -E "error"
/tmp/spt 1 144 | tail -8 | cut -c1-300 > gpurun_out/r4_flushfix/tail_steps.txt; cat gpurun_out/r4_flushfix/tail_steps.txt
bash scripts/jobs/r4_step.sh
cp gpurun_out/r4_step/* gpurun_out/r4_flushfix/
