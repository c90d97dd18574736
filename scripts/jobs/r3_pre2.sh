# Round 3: pre-processing kernels after the rewrite (parity, timings, rocprof), then timing experiments on the one-pass kernel
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r3_pre2
cd $R
timeout 900 python -m pytest tests/test_gpu_preprocessing.py -x -q > gpurun_out/r3_pre2/pytest_pre.log 2>&1; tail -4 gpurun_out/r3_pre2/pytest_pre.log
timeout 600 python -m pytest tests/test_gpu_fuzz_parity.py -x -q -k "preprocess or blur or quantile or rowfilter" > gpurun_out/r3_pre2/pytest_fuzz.log 2>&1; tail -3 gpurun_out/r3_pre2/pytest_fuzz.log
bash scripts/jobs/r3_pre_prof.sh
cp -r gpurun_out/r3_pre gpurun_out/r3_pre2/prof
# timing experiments (hooks compiled in on this box only)
PXSOM_ACC_EXPERIMENT=1 python ark_analysis_amd/_build.py > /dev/null 2>&1
for m in 0 1 2 3; do
  PXSOM_ACC_EXPERIMENT=1 PXSOM_ACC_EXP=$m python bench.py --steps 5 --warmup 2 --no-pmc --no-cpu-baseline --no-online --one-pass 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('exp $m',d['phases_ms'])" | tee -a gpurun_out/r3_pre2/acc_exp.txt
done
