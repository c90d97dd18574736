# flush of per-workgroup tables into one statistics buffer vs replicas (scripts/ubench/flush_replicas.hip) -> gpurun_out/r4_flush.txt
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value scripts/ubench/flush_replicas.hip -o /tmp/flr && timeout 120 /tmp/flr 2>&1 | tee gpurun_out/r4_flush.txt
