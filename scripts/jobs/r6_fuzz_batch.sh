# Round 6: the batch-training tests of the randomised sweep only (route changes of the training loop)
R=$GRAFT_REPO_ROOT
cd $R; O=gpurun_out/r6_fuzz_batch; mkdir -p $O; rm -f $O/summary.txt
for seed in 91 92 93; do PXSOM_FUZZ_CASES=${CASES:-1500} PXSOM_FUZZ_SEED=$seed timeout 2400 python -m pytest tests/test_gpu_fuzz_parity.py -q -x -k "batch" > $O/seed_$seed.log 2>&1
  echo "seed $seed, ${CASES:-1500} cases per test: $(grep -E 'passed|failed' $O/seed_$seed.log | tail -1)" | tee -a $O/summary.txt; grep -E "^E  " $O/seed_$seed.log | head -5; done
