"""bench.py's N > 1 code on a one-GPU box: two ranks launched exactly as the driver launches them
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 ... bench.py --gpus 2 ...`), both on device 0 over a gloo
group (PXSOM_BENCH_DRY_RANKS=1: RCCL will not put two ranks on one GPU).  What it proves: the script's distributed code
runs -- rank-sharded rows, the agreed kernel route, the per-step exchange (through torch.distributed, and through the
library's own communicator over tests/mock_rccl), the table all-reduce, MAX-over-ranks timing, per-rank phases -- and
prints one well-formed line.  What it cannot prove: anything about speed.  `-m gpu` only."""
import json
import os
import socket
import subprocess
import sys

import pytest

from tests.test_gpu_exchange import _mock_library

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("exchange", ["torch.distributed", "in-library", "p2p", "fused"])
@pytest.mark.parametrize("config", ["cfg2", "cfg4"])
def test_two_rank_bench_line(gpu, tmp_path, exchange, config):
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", PXSOM_BENCH_DRY_RANKS="1")
    if exchange == "in-library":
        env.update(PXSOM_NATIVE_EXCHANGE="force", PXSOM_RCCL_LIBRARY=_mock_library(tmp_path))
    elif exchange in ("p2p", "fused"):   # the library's peer-to-peer exchange: real on one device (HIP IPC between the two processes);
        env.update(PXSOM_EXCHANGE=exchange)   # "fused": inside the 10 x 10 step's own launch (config 2; config 4 keeps the one-launch exchange)
    else:
        env.update(PXSOM_NATIVE_EXCHANGE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--config", config, "--fovs-per-gpu", "1", "--no-pmc", "--no-cpu-baseline", "--no-online", "--no-operating-range"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert res.returncode == 0 and len(lines) == 1, res.stdout[-2000:] + res.stderr[-4000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1 and line["value"] > 0
    assert line["config"]["name"] == config and "dry_run" in line["config"]
    assert line["config"]["exchange"].startswith("torch.distributed" if exchange == "torch.distributed" else "in-library")
    per_rank = line["phases_ms"]["per_rank"]
    assert len(per_rank["train_batch"]) == 2 and all(v > 0 for v in per_rank["train_batch"])
    assert per_rank["exchange_us_per_step"] > 0 and per_rank["exchange_ms_per_pass"] > 0     # the exchange timed on its own
    if exchange in ("p2p", "fused"):
        assert per_rank["exchange_route"] == "P2PComm" and per_rank["exchange_us_per_step_p2p"] is None
    else:                        # ... and the peer-to-peer route beside whichever one the pass used
        assert per_rank["exchange_us_per_step_p2p"] > 0
    assert line["scaling"] == ("weak" if config == "cfg2" else "strong")


@pytest.mark.parametrize("config,exchange", [("cfg2", "torch.distributed"), ("cfg3", "in-library"), ("cfg4", "torch.distributed"),
                                             ("cfg2", "fused")])
def test_eight_rank_bench_line(gpu, tmp_path, config, exchange):
    """The driver's `--gpus 8` launch with all eight ranks on device 0 (gloo; one FOV resp. 125 000 cells per rank): ONE line,
    every per-rank list eight long, the replicas' codebooks equal bit for bit after the timed passes, the kernel route agreed by
    all ranks (config 4 at eight ranks is the 125 K-cell shard: wide one-launch steps where a step holds at most 16 K rows, launch
    per phase beyond -- the same bits either way), the exchange route the line names is the one asked for."""
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", PXSOM_BENCH_DRY_RANKS="1")
    if exchange == "torch.distributed":
        env.update(PXSOM_NATIVE_EXCHANGE="0")
    elif exchange == "in-library":
        env.update(PXSOM_NATIVE_EXCHANGE="force", PXSOM_RCCL_LIBRARY=_mock_library(tmp_path), PXSOM_EXCHANGE="rccl")
    else:
        # Eight processes on ONE device cannot all hold a spinning exchange kernel at once (the device time-slices them): the
        # peer-to-peer route is made, fails its first checked exchange within its bounded wait, and is dropped by all ranks
        # together -- the line must come out either way and say which route it took and why
        env.update(PXSOM_EXCHANGE=exchange)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
           "--config", config, "--fovs-per-gpu", "1", "--no-pmc", "--no-cpu-baseline", "--no-online", "--no-operating-range"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500, cwd=str(tmp_path))
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert res.returncode == 0 and len(lines) == 1, res.stdout[-2000:] + res.stderr[-4000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["value"] > 0 and line["config"]["name"] == config
    per_rank = line["phases_ms"]["per_rank"]
    assert len(per_rank["train_batch"]) == 8 and len(per_rank["assign_and_mean_table"]) == 8
    assert all(v > 0 for v in per_rank["train_batch"])
    assert per_rank["codebooks_equal"] is True
    agreed = per_rank["kernel_route_agreement"]
    assert agreed is not None and agreed["unfused_now"] is False
    if config == "cfg4":
        assert agreed["all_fused"] is False and agreed["any_fused"] is False      # no rank has the 10 x 10 x <= 32 step: nothing forced
        assert line["config"]["rows_per_gpu"] == 125_000 and line["scaling"] == "strong"
    else:
        assert agreed["all_fused"] is True
    decision = per_rank["exchange_decision"]
    if exchange == "fused":
        assert decision["route_taken"] in ("p2p-fused", "torch.distributed")
        if decision["route_taken"] == "torch.distributed":
            assert "unavailable" in decision["p2p"] and per_rank["exchange_route"] == "torch.distributed"
    else:
        assert per_rank["exchange_route"] == ("torch.distributed" if exchange == "torch.distributed" else "RankComm")
        assert decision["route_taken"] == ("torch.distributed" if exchange == "torch.distributed" else "rccl")
        assert per_rank["exchange_fused"] is False
