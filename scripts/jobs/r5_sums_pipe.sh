# Round 5: atomic sums kernel with the next batch of loads requested before the current one goes into the table: parity, wide probe, config 4 / 5 lines
R=$GRAFT_REPO_ROOT
cd $R; O=gpurun_out/r5_sums_pipe; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_som_kernels.py tests/test_gpu_wide_rows.py tests/test_gpu_fuzz_parity.py tests/test_gpu_schedule.py -m gpu -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
PXSOM_FUZZ_DTYPE=f16 PXSOM_FUZZ_CASES=500 timeout 900 python -m pytest tests/test_gpu_fuzz_parity.py -m gpu -x -q -k "assign_and_sums" 2>&1 | tail -1
python scripts/debug/label_coherence_probe_wide.py 2>&1 | grep -v amdgpu.ids | tee $O/probe_wide.txt
for c in cfg5 cfg4; do python bench.py --config $c --steps 3 --warmup 1 --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$c', d['value'], d['ms_per_step'], d['phases_ms'])"; done | tee $O/lines.txt
