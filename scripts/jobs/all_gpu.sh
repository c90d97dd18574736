# The whole -m gpu suite, smoke(), one default bench line -> gpurun_out/all_gpu/
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/all_gpu
cd $R
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/all_gpu/pytest.log 2>&1; tail -5 gpurun_out/all_gpu/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/all_gpu/smoke.log 2>&1; tail -2 gpurun_out/all_gpu/smoke.log
python bench.py > gpurun_out/all_gpu/bench_default.json 2> gpurun_out/all_gpu/bench_default.err; python -c "
import json;d=json.loads(open('gpurun_out/all_gpu/bench_default.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['phases_ms'],d.get('roofline'),d.get('batch_train'))"
