# Round 5: the sweep repeated on the final tree (scan threshold 47, queue trips outside the decision)
R=$GRAFT_REPO_ROOT
cd $R; O=gpurun_out/r5_fuzz2; mkdir -p $O
run() { PXSOM_FUZZ_CASES=$1 PXSOM_FUZZ_SEED=$2 timeout 2400 python -m pytest tests/test_gpu_fuzz_parity.py -q -x > $O/seed_$2.log 2>&1
  echo "seed $2, $1 cases per test: $(grep -E 'passed|failed' $O/seed_$2.log | tail -1)" | tee -a $O/summary.txt; grep -E "^E  " $O/seed_$2.log | head -5; }
for seed in 71 72 73; do run 600 $seed; done
run 2500 74
for dt in f32 f16 f64; do PXSOM_FUZZ_DTYPE=$dt PXSOM_FUZZ_CASES=800 PXSOM_FUZZ_SEED=75 timeout 2400 python -m pytest tests/test_gpu_fuzz_parity.py -q -x -k "assign_and_sums" > $O/seed_75_$dt.log 2>&1
  echo "seed 75, 800 cases, $dt rows only, assign + sums + one-pass (a third of the cases in runs of equal labels): $(grep -E 'passed|failed' $O/seed_75_$dt.log | tail -1)" | tee -a $O/summary.txt; done
