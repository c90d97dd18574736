# Round 6: rows listed per fused training step (debug counter build) -> gpurun_out/r6_listed/
R=$GRAFT_REPO_ROOT
cd $R; O=gpurun_out/r6_listed; mkdir -p $O
python scripts/dev/listed_per_step.py ark_analysis_amd/variants/cnt.so --steps 1 --warmup 0 --no-cpu-baseline --no-online --no-pmc --no-operating-range 2>/dev/null | tail -120 > $O/listed.txt
tail -60 $O/listed.txt
