# Round 5, item 1: libm-free gain (batch_gain / orc_batch_gain).  The case probe, the tests that compare whole runs and
# single updates with the oracle (now bit for bit), a fuzz sweep of the two batch tests, one default bench line.
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r5_gain
python scripts/debug/tail_case_probe.py > gpurun_out/r5_gain/tail_case_probe.txt 2>&1; tail -12 gpurun_out/r5_gain/tail_case_probe.txt
timeout 2400 python -m pytest tests/test_gpu_schedule.py tests/test_gpu_fuzz_parity.py tests/test_pipeline_dropin.py tests/test_gpu_som_kernels.py tests/test_gpu_wide_rows.py -m gpu -x -q > gpurun_out/r5_gain/pytest.log 2>&1; tail -15 gpurun_out/r5_gain/pytest.log
PXSOM_FUZZ_CASES=150 timeout 1500 python -m pytest tests/test_gpu_fuzz_parity.py -m gpu -x -q -k "batch" > gpurun_out/r5_gain/fuzz150.log 2>&1; tail -8 gpurun_out/r5_gain/fuzz150.log
python bench.py > gpurun_out/r5_gain/bench_default.json 2> gpurun_out/r5_gain/bench_default.err; python -c "
import json;d=json.loads(open('gpurun_out/r5_gain/bench_default.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['phases_ms'],d.get('roofline'),d.get('batch_train'))"
