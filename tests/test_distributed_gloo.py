"""The FOV-sharded batch trainer on two CPU processes (gloo): sharding, the per-step packed
all-reduce and the identical-on-every-rank codebook, checked against the single-process oracle.
The compute kernels are the oracle here (test infrastructure) -- this exercises the host logic
of ark_analysis_amd.distributed, which is backend-agnostic; the HIP kernels are covered by the
`-m gpu` tests."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ark_analysis_amd.distributed import (BatchSOMTrainer, allreduce_cluster_tables, batch_schedule,
                                          broadcast_codebook)


from tests.oracle_backend import OracleKernels  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, shards, w0, xdim, ydim, m, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x = torch.from_numpy(shards[rank])
    w = torch.from_numpy(w0.copy()) if rank == 0 else torch.zeros(w0.shape, dtype=torch.float64)
    broadcast_codebook(w, 0)
    trainer = BatchSOMTrainer(xdim, ydim, x.shape[1], "cpu", batch_steps=m, kernels=OracleKernels())
    trainer.train(x, w, num_passes=2)
    # K8 across ranks: per-cluster tables of the final labels
    from tests import oracle_binding as ob
    xn = np.ascontiguousarray(x.numpy(), dtype=np.float64)
    lab, _ = ob.map_data_to_nodes(w.numpy(), xn)
    s_np, cnt_np = ob.cluster_sums(xn, lab, xdim * ydim)
    sums = torch.from_numpy(s_np).clone()
    counts = torch.from_numpy(cnt_np.astype(np.int64))
    allreduce_cluster_tables(sums, counts)
    gathered = [torch.zeros_like(w) for _ in range(world)]
    dist.all_gather(gathered, w)
    if rank == 0:
        np.savez(out_path, w=w.numpy(), same=np.array([bool(torch.equal(g, w)) for g in gathered]),
                 sums=sums.numpy(), counts=counts.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_two_rank_batch_training_matches_single_process_oracle(oracle, tmp_path, world):
    """FOV-sharded batch training over `world` ranks (gloo, CPU; 8 = the node the scaling bench runs on): replicas equal, the
    codebook that of ONE process on the united rows, the all-reduced cluster tables those of the united rows."""
    from ark_analysis_amd.flowsom import default_radius_range
    xdim = ydim = 5
    k, c, m, n_local = 25, 6, 8, 400
    rs = np.random.RandomState(0)
    shards = [rs.gamma(0.8, 0.4, size=(n_local, c)) for _ in range(world)]
    w0 = shards[0][rs.choice(n_local, k, replace=False)].copy()
    out = str(tmp_path / "rank0.npz")
    mp.spawn(_worker, args=(world, _free_port(), shards, w0, xdim, ydim, m, out), nprocs=world, join=True)
    res = np.load(out)
    assert res["same"].all() and len(res["same"]) == world, "codebook differs between ranks"
    # single-process equivalent: interleave the shards in blocks of m rows so that global row i % m
    # selects exactly the union of all ranks' local mini-batch (i % m)
    blocks = []
    for j in range(n_local // m):
        for r in range(world):
            blocks.append(shards[r][j * m:(j + 1) * m])
    g = np.concatenate(blocks)
    want = oracle.som_batch(g, w0, xdim, ydim, 2, (0.05, 0.01), default_radius_range(xdim, ydim), m)
    np.testing.assert_allclose(res["w"], want, rtol=1e-10, atol=0)
    lab, _ = oracle.map_data_to_nodes(want, g)
    s, cnt = oracle.cluster_sums(g, lab, k)
    np.testing.assert_array_equal(res["counts"], cnt)
    np.testing.assert_allclose(res["sums"], s, rtol=1e-9, atol=1e-12)


def test_batch_schedule_endpoints():
    assert batch_schedule(0, 64, (0.05, 0.01), (6.0, 0.0)) == (6.0, 0.05)
    thr, alpha = batch_schedule(63, 64, (0.05, 0.01), (6.0, 0.0))
    assert thr == 0.5 and abs(alpha - (0.05 - 0.04 * 63 / 64)) < 1e-15   # below 1 -> BMU only
    assert batch_schedule(32, 64, (0.05, 0.01), (6.0, 0.0))[0] == 3.0
