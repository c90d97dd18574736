# bench.py launched the way the driver launches N > 1, with one rank: backend nccl (RCCL), in-library communicator, the whole distributed code path
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5_torchrun1
for ex in rccl p2p fused; do
PXSOM_EXCHANGE=$ex timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-online --no-pmc --no-operating-range > gpurun_out/r5_torchrun1/line_$ex.json 2> gpurun_out/r5_torchrun1/err_$ex.txt
echo "== $ex rc $?"; tail -1 gpurun_out/r5_torchrun1/line_$ex.json | python -c "
import json,sys;d=json.loads(sys.stdin.read());print(d['value'], d['ms_per_step'], d['n_gpus'], d['config']['exchange'], d['phases_ms'].get('per_rank'))" || tail -5 gpurun_out/r5_torchrun1/err_$ex.txt
done
