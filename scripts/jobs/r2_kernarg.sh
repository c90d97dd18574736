for v in 0 1; do
  echo "HIP_FORCE_DEV_KERNARG=$v"
  HIP_FORCE_DEV_KERNARG=$v PXSOM_STEP_TPW=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-online | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"phases_ms".*"mean_table": [0-9.]*' | tr '\n' ' '; echo
done
C=ark_analysis_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -Iinclude -I$C scripts/ubench/step_phase_timing.hip $C/pxsom_api.hip $C/pxsom_assign_filter.hip $C/pxsom_assign_filter_acc.hip -ffinite-math-only -o /tmp/spt 2>&1 | grep -E "error"
for v in 0 1; do echo "KERNARG $v"; HIP_FORCE_DEV_KERNARG=$v /tmp/spt 1 | grep "^step 20\|^rep 1"; done
