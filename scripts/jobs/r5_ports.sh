# Round 5: what takes the MFMA / VALU overlap away -- operand count of the vector instructions, distinct MFMA operands, AccVGPR operands?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5_overlap_shapes
hipcc --offload-arch=gfx950 -O3 -o /tmp/mvp scripts/ubench/mfma_valu_ports.hip && /tmp/mvp | tee gpurun_out/r5_overlap_shapes/ports.txt
