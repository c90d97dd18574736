# XCD-local synchronisation probe (scripts/ubench/xcd_local_sync.hip) -> gpurun_out/r4_xcd_sync.txt
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value scripts/ubench/xcd_local_sync.hip -o /tmp/xls && timeout 120 /tmp/xls 2>&1 | tee gpurun_out/r4_xcd_sync.txt
