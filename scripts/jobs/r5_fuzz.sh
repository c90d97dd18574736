# Round 5: the randomised parity sweep on the final tree (new tolerance, row-axis sums on coherent tiles; the sums test now also reorders a
# third of its cases into runs of equal labels) -> gpurun_out/r5_fuzz/
R=$GRAFT_REPO_ROOT
cd $R; O=gpurun_out/r5_fuzz; mkdir -p $O
run() { PXSOM_FUZZ_CASES=$1 PXSOM_FUZZ_SEED=$2 timeout 2400 python -m pytest tests/test_gpu_fuzz_parity.py -q -x > $O/seed_$2.log 2>&1
  echo "seed $2, $1 cases per test: $(grep -E 'passed|failed' $O/seed_$2.log | tail -1)" | tee -a $O/summary.txt; grep -E "^E  " $O/seed_$2.log | head -5; }
for seed in 61 62 63 64; do run 600 $seed; done
for seed in 65 66; do run 2500 $seed; done
for dt in f16 f64; do PXSOM_FUZZ_DTYPE=$dt PXSOM_FUZZ_CASES=1000 PXSOM_FUZZ_SEED=67 timeout 2400 python -m pytest tests/test_gpu_fuzz_parity.py -q -x > $O/seed_67_$dt.log 2>&1
  echo "seed 67, 1000 cases per test, $dt rows only: $(grep -E 'passed|failed' $O/seed_67_$dt.log | tail -1)" | tee -a $O/summary.txt; done
