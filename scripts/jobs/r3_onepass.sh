# Round 3: one-pass labels + mean table with fixed-point workgroup tables -- parity, bench with and without, kernel trace
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r3_onepass
cd $R
timeout 600 python -m pytest tests/test_gpu_som_kernels.py -x -q -k "one_pass or batch_train_steps_fused" > gpurun_out/r3_onepass/pytest.log 2>&1; tail -4 gpurun_out/r3_onepass/pytest.log
python bench.py --steps 5 --warmup 2 --no-pmc --no-cpu-baseline --no-online --one-pass > gpurun_out/r3_onepass/bench_onepass.json 2>gpurun_out/r3_onepass/err.txt; python -c "
import json;d=json.loads(open('gpurun_out/r3_onepass/bench_onepass.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['phases_ms'])"
PXSOM_SUMS_F64=1 python bench.py --steps 5 --warmup 2 --no-pmc --no-cpu-baseline --no-online --one-pass > gpurun_out/r3_onepass/bench_onepass_f64.json 2>/dev/null; python -c "
import json;d=json.loads(open('gpurun_out/r3_onepass/bench_onepass_f64.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['phases_ms'])"
bash scripts/jobs/r3_trace.sh
