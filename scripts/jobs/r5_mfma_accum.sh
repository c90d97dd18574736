# Round 5: how does v_mfma_f32_16x16x32_f16 round?  (scripts/ubench/mfma_accum.hip on the cases of scripts/debug/mfma_accum_cases.py)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_mfma_accum; mkdir -p $O
python scripts/debug/mfma_accum_cases.py gen $O/in.bin
hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_accum scripts/ubench/mfma_accum.hip && /tmp/mfma_accum $O/in.bin $O/out.bin
python scripts/debug/mfma_accum_cases.py fit $O/in.bin $O/out.bin 2>&1 | tee $O/fit.txt
