python -m tests.tools.robustness_sweep 2>&1 | grep -v amdgpu.ids | tee gpurun_out/robustness_sweep.txt
