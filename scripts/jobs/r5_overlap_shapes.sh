# Round 5: do matrix and vector instructions of different waves on one SIMD overlap, by MFMA shape? (scripts/ubench/mfma_valu_overlap.hip)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5_overlap_shapes
hipcc --offload-arch=gfx950 -O3 -o /tmp/mvo scripts/ubench/mfma_valu_overlap.hip && /tmp/mvo | tee gpurun_out/r5_overlap_shapes/overlap.txt
