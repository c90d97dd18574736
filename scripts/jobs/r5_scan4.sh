# scan decision not taken on / after queue trips: parity, LDS instruction count on the synthetic order, interleaved default lines
R=$GRAFT_REPO_ROOT
cd $R; O=gpurun_out/r5_scan; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_som_kernels.py tests/test_gpu_fuzz_parity.py -m gpu -x -q > $O/pytest4.log 2>&1; tail -2 $O/pytest4.log
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmc_h && rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d /tmp/pmc_h -o pmc -- python $R/scripts/debug/label_coherence_pmc_target.py generated > /dev/null 2>&1)
python scripts/prof_summarize.py /tmp/pmc_h $O/pmc_lds_head_generated.txt > /dev/null; grep "bmu_filter_fastIfLi6ELi7ELi1ELi0ELb1ELb1ELb1E" $O/pmc_lds_head_generated.txt | cut -c1-20,96-200
line() { PXSOM_HIPCC_EXTRA="$2" python bench.py --no-cpu-baseline --no-online --no-pmc --no-operating-range 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$1 cfg2', d['value'], d['ms_per_step'], d['phases_ms'])"; }
{ line scan ""; line scan ""
PXSOM_HIPCC_EXTRA="-DPXSOM_ADD_SCAN=0" python -c "from ark_analysis_amd import _build; _build.build(force=True)" > $O/build_off.log 2>&1 || tail -5 $O/build_off.log
line off "-DPXSOM_ADD_SCAN=0"; line off "-DPXSOM_ADD_SCAN=0"
python -c "from ark_analysis_amd import _build; _build.build(force=True)" > $O/build_on.log 2>&1
line scan ""; line scan ""; python scripts/debug/label_coherence_probe.py 2>&1 | grep -v amdgpu.ids | head -4; } | tee $O/bench_ab4.txt
