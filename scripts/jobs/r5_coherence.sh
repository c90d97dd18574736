# Round 5: the one-pass labels + mean-table kernel on rows whose neighbours share their label (real images) -- scripts/debug/label_coherence_probe.py
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5_coherence
python scripts/debug/label_coherence_probe.py 2>&1 | tee gpurun_out/r5_coherence/probe.txt
