python scripts/robustness_sweep.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/robustness_sweep.txt
