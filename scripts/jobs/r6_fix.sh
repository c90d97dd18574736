# Round 6: one-pass kernel with one table address per tile (row stride 4 CPL + 1, count in the slot past the row's end): parity + bench lines
R=$GRAFT_REPO_ROOT
cd $R; O=gpurun_out/r6_fix; mkdir -p $O
timeout 1700 python -m pytest tests/test_gpu_som_kernels.py tests/test_gpu_fuzz_parity.py tests/test_gpu_alternate_routes.py -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
line() { python bench.py --no-cpu-baseline --no-online --no-pmc 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$1', d['value'], d['ms_per_step'], d['phases_ms'], d['roofline'], d.get('operating_range'))"; }
{ line head; line head; line head; } | tee $O/bench.txt
