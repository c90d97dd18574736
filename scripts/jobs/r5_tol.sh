# Round 5: the filter's accumulation term after the measured behaviour of the matrix unit (group additions + cuts, pxsom_assign.h) against the
# slot-wise term of rounds 1 - 4 (-DPXSOM_TOL_SLOTWISE=1): whole GPU suite + fuzz on the new bound, then same-box lines of configs 2, 4, 5
R=$GRAFT_REPO_ROOT
cd $R; O=gpurun_out/r5_tol; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for dt in f32 f16 f64; do PXSOM_FUZZ_DTYPE=$dt PXSOM_FUZZ_CASES=600 timeout 1200 python -m pytest tests/test_gpu_fuzz_parity.py -m gpu -x -q 2>&1 | tail -1; done
line() { PXSOM_HIPCC_EXTRA="$3" python bench.py --config $2 --steps 5 --warmup 2 --no-pmc --no-cpu-baseline --no-online 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$1 $2', d['value'], d['ms_per_step'], d['phases_ms'], (d.get('operating_range') or {}).get('codebook = data rows'))"; }
{
for c in cfg4 cfg2 cfg5 cfg3; do line groups $c ""; done
PXSOM_HIPCC_EXTRA="-DPXSOM_TOL_SLOTWISE=1" python -c "from ark_analysis_amd import _build; _build.build(force=True)" > $O/build_old.log 2>&1 || tail -5 $O/build_old.log
for c in cfg4 cfg2 cfg5 cfg3; do line slotwise $c "-DPXSOM_TOL_SLOTWISE=1"; done
} 2>&1 | tee $O/summary.txt
