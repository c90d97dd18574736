# Randomised parity sweep with N cases per test and several seeds -> gpurun_out/r3_fuzz/
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r3_fuzz
N=${1:-300}
for seed in ${SEEDS:-31 32 33}; do
  PXSOM_FUZZ_CASES=$N PXSOM_FUZZ_SEED=$seed timeout 2400 python -m pytest tests/test_gpu_fuzz_parity.py -q -x > gpurun_out/r3_fuzz/seed_$seed.log 2>&1
  echo "seed $seed: $(grep -E 'passed|failed' gpurun_out/r3_fuzz/seed_$seed.log | tail -1)" | tee -a gpurun_out/r3_fuzz/summary.txt
  grep -E "^E  " gpurun_out/r3_fuzz/seed_$seed.log | head -5
done
