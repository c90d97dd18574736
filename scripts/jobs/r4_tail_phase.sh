# phase stamps inside the persistent tail kernel, slot-load experiments -> gpurun_out/r4_tail_phase.txt
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
C=ark_analysis_amd/csrc
rm -f gpurun_out/r4_tail_phase.txt
for m in 0 1 2 4; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -Wno-unused-value -DPXSOM_TAIL_SLOT_MODE=$m -Iinclude -I$C scripts/ubench/tail_phase_timing.hip $C/pxsom_api.hip -o /tmp/tpt_$m 2>&1 | grep -E "error"
  echo "=== slot mode $m" >> gpurun_out/r4_tail_phase.txt
  (timeout 120 /tmp/tpt_$m 8) 2>&1 | grep -v "rep 0" | grep -A 40 "rep 2" >> gpurun_out/r4_tail_phase.txt
done
cat gpurun_out/r4_tail_phase.txt
