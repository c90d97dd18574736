# Round 5: the two-pass sums kernel (pairs form) on groups of rows that all carry one label: parity + probe
R=$GRAFT_REPO_ROOT
cd $R; O=gpurun_out/r5_pairs; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_som_kernels.py tests/test_gpu_fuzz_parity.py tests/test_pipeline_dropin.py -m gpu -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
python scripts/debug/label_coherence_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/probe.txt
python bench.py --two-pass --steps 5 --warmup 2 --no-cpu-baseline --no-online --no-operating-range --no-pmc 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('two-pass cfg2', d['value'], d['ms_per_step'], d['phases_ms'])"
