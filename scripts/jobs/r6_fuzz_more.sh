# Round 6: more seeds of the randomised parity sweep (SEEDS="101 102", CASES=2500)
R=$GRAFT_REPO_ROOT
cd $R; O=gpurun_out/r6_fuzz_more; mkdir -p $O
for seed in ${SEEDS:-101 102}; do PXSOM_FUZZ_CASES=${CASES:-2500} PXSOM_FUZZ_SEED=$seed timeout 2400 python -m pytest tests/test_gpu_fuzz_parity.py -q > $O/seed_$seed.log 2>&1
  echo "seed $seed, ${CASES:-2500} cases per test: $(grep -E 'passed|failed' $O/seed_$seed.log | tail -1)" | tee -a $O/summary.txt; grep -E "AssertionError: case" $O/seed_$seed.log | cut -c1-220 | head -8; done
