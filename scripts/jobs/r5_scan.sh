# Round 5: one-pass labels + mean-table kernel, tiles of label-coherent rows summed along the row axis before they touch the table (PXSOM_ADD_SCAN)
R=$GRAFT_REPO_ROOT
cd $R; O=gpurun_out/r5_scan; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_som_kernels.py tests/test_gpu_fuzz_parity.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python scripts/debug/label_coherence_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/probe_scan52.txt
python bench.py --no-cpu-baseline --no-online --no-pmc 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('scan52 cfg2', d['value'], d['ms_per_step'], d['phases_ms'])" | tee -a $O/probe_scan52.txt
for v in 44 32; do
PXSOM_HIPCC_EXTRA="-DPXSOM_ADD_SCAN_MIN=$v" python -c "from ark_analysis_amd import _build; _build.build(force=True)" > $O/build_$v.log 2>&1 || tail -5 $O/build_$v.log
PXSOM_HIPCC_EXTRA="-DPXSOM_ADD_SCAN_MIN=$v" python scripts/debug/label_coherence_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/probe_scan$v.txt
done
PXSOM_HIPCC_EXTRA="-DPXSOM_ADD_SCAN=0" python -c "from ark_analysis_amd import _build; _build.build(force=True)" > $O/build_off.log 2>&1 || tail -5 $O/build_off.log
PXSOM_HIPCC_EXTRA="-DPXSOM_ADD_SCAN=0" python scripts/debug/label_coherence_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/probe_off.txt
PXSOM_HIPCC_EXTRA="-DPXSOM_ADD_SCAN=0" python bench.py --no-cpu-baseline --no-online --no-pmc 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('off cfg2', d['value'], d['ms_per_step'], d['phases_ms'])" | tee -a $O/probe_off.txt
