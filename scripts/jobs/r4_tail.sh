# persistent tail: parity tests, phase stamps, timing of the default bench (with / without the tail) -> gpurun_out/r4_tail/
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r4_tail
cd $R
timeout 900 python -m pytest tests/test_gpu_schedule.py -m gpu -x -q -k "persistent_tail or default_schedule_run or scheduled_steps" > gpurun_out/r4_tail/pytest.log 2>&1; tail -5 gpurun_out/r4_tail/pytest.log
C=ark_analysis_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -Wno-unused-value -Iinclude -I$C scripts/ubench/tail_phase_timing.hip $C/pxsom_api.hip -o /tmp/tpt 2>&1 | grep -E "error"
(timeout 120 /tmp/tpt 8; timeout 120 /tmp/tpt 15) 2>&1 | grep -v "rep 0" > gpurun_out/r4_tail/phase.txt; cat gpurun_out/r4_tail/phase.txt
for v in 0 1; do
  if [ $v = 0 ]; then export PXSOM_TRAIN_TAIL=1; else unset PXSOM_TRAIN_TAIL; fi
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-online --no-pmc --no-operating-range > gpurun_out/r4_tail/bench_notail$v.json 2> gpurun_out/r4_tail/bench_notail$v.err
  python -c "
import json;d=json.loads(open('gpurun_out/r4_tail/bench_notail$v.json').read().strip().splitlines()[-1]);print('no_tail=$v',d['value'],d['ms_per_step'],d['phases_ms'],d.get('batch_train'))"
done
