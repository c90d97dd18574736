# Round 5: atomic sums kernel (config 5's mean table: binary16 x 40 channels, 400 clusters), 16-byte loads in flight per thread: 4 (HEAD) / 8 / 6
R=$GRAFT_REPO_ROOT
cd $R; O=gpurun_out/r5_sums_depth; mkdir -p $O
for d in 4 8 6; do
  PXSOM_HIPCC_EXTRA="-DPXSOM_SUMS_IN_FLIGHT=$d" python -c "from ark_analysis_amd import _build; _build.build(force=True)" > $O/build_$d.log 2>&1 || { tail -5 $O/build_$d.log; continue; }
  echo "== in flight $d"; PXSOM_HIPCC_EXTRA="-DPXSOM_SUMS_IN_FLIGHT=$d" python scripts/debug/label_coherence_probe_wide.py 2>&1 | grep -v amdgpu.ids | grep "rows, \|sorted"
done 2>&1 | tee $O/summary.txt
