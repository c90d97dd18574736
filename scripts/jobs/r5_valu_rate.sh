# Issue rates of single VALU instructions (scripts/ubench/valu_rate.hip): what the fixed-point conversion of the table adds costs
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/r5_valu_rate
hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate scripts/ubench/valu_rate.hip && /tmp/valu_rate | tee gpurun_out/r5_valu_rate/valu_rate.txt
