# Round 5, item 5: the rule's exchange inside the step launches (PXSOM_EXCHANGE=fused), two processes on one device
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r5_fused; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_exchange.py -m gpu -x -q > $O/pytest_exchange.log 2>&1; tail -4 $O/pytest_exchange.log
timeout 2400 python -m pytest tests/test_gpu_bench_multirank.py tests/test_gpu_distributed.py tests/test_gpu_schedule.py -m gpu -x -q > $O/pytest_more.log 2>&1; tail -3 $O/pytest_more.log
# the pass of a 2-rank job on one device, exchange per route (timings of two processes sharing one GPU: relative only)
for ex in p2p fused p2p fused; do
  PXSOM_EXCHANGE=$ex PXSOM_BENCH_DRY_RANKS=1 HSA_ENABLE_IPC_MODE_LEGACY=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 2 --steps 10 --warmup 3 --no-pmc --no-cpu-baseline --no-online --no-operating-range 2>/dev/null | grep "^{" | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$ex', d['ms_per_step'], d['phases_ms']['per_rank']['train_batch'], d['phases_ms']['per_rank'].get('exchange_us_per_step'), d['config']['exchange'][:60])"
done | tee $O/two_rank_pass.txt
