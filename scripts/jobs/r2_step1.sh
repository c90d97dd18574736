timeout 900 python -m pytest tests/test_gpu_som_kernels.py -x -q -k "batch" 2>&1 | tail -15
for tpw in 1 2; do
  PXSOM_STEP_TPW=$tpw python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-online | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"phases_ms".*"mean_table": [0-9.]*' | tr '\n' ' '; echo " tpw=$tpw"
done
bash scripts/jobs/r2_phase.sh
