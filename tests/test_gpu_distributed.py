"""Two ranks on the GPU box: the FOV-sharded batch trainer with the real HIP kernels on both ranks
(both processes share cuda:0; the collective goes through gloo, which copies device tensors through
the host -- RCCL cannot put two ranks on one device).  Checks rank agreement and parity with the
single-process oracle.  `-m gpu` only."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, shards, w0, xdim, ydim, m, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from ark_analysis_amd import som_device
    from ark_analysis_amd.distributed import BatchSOMTrainer, allreduce_cluster_tables, broadcast_codebook
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    x = torch.from_numpy(shards[rank]).to(dev)
    w = torch.from_numpy(w0.copy()).to(dev) if rank == 0 else torch.zeros(w0.shape, dtype=torch.float64, device=dev)
    broadcast_codebook(w, 0)
    trainer = BatchSOMTrainer(xdim, ydim, x.shape[1], dev, batch_steps=m)   # HipKernels
    trainer.train(x, w, num_passes=1)
    labels, _ = som_device.assign(x, w)
    sums, counts = som_device.cluster_sums(x, labels, xdim * ydim)
    allreduce_cluster_tables(sums, counts)
    gathered = [torch.zeros_like(w) for _ in range(world)]
    dist.all_gather(gathered, w)
    if rank == 0:
        np.savez(out_path, w=w.cpu().numpy(), same=np.array([bool(torch.equal(g, w)) for g in gathered]),
                 sums=sums.cpu().numpy(), counts=counts.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_hip_kernels_match_oracle(gpu, oracle, tmp_path):
    from ark_analysis_amd import synth
    from ark_analysis_amd.flowsom import default_radius_range
    xdim = ydim = 10
    k, c, m, n_local = 100, 22, 8, 8000
    shards = [synth.make_fov_numpy(n_local, c, seed=50 + r, dtype=np.float32) for r in range(2)]
    rs = np.random.RandomState(0)
    w0 = shards[0][rs.choice(n_local, k, replace=False)].astype(np.float64)
    out = str(tmp_path / "rank0.npz")
    mp.spawn(_worker, args=(2, _free_port(), shards, w0, xdim, ydim, m, out), nprocs=2, join=True)
    res = np.load(out)
    assert res["same"].all(), "codebook differs between ranks"
    blocks = []
    for j in range(n_local // m):
        for r in range(2):
            blocks.append(shards[r][j * m:(j + 1) * m])
    g = np.concatenate(blocks).astype(np.float64)
    want = oracle.som_batch(g, w0, xdim, ydim, 1, (0.05, 0.01), default_radius_range(xdim, ydim), m)
    np.testing.assert_allclose(res["w"], want, rtol=1e-9, atol=0)
    lab, _ = oracle.map_data_to_nodes(res["w"], g)
    s, cnt = oracle.cluster_sums(g, lab, k)
    np.testing.assert_array_equal(res["counts"], cnt)
    np.testing.assert_allclose(res["sums"], s, rtol=1e-9, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(10, 10, 22), (20, 20, 40), (5, 7, 3)])
def test_empty_and_tiny_shards(shape):
    """A rank that was dealt no rows (fewer FOVs than ranks) or fewer rows than mini-batch steps: the step loop must
    run (the other ranks wait in the exchange), contribute nothing and leave the codebook finite."""
    import torch
    from ark_analysis_amd import som_device
    from ark_analysis_amd.distributed import BatchSOMTrainer
    xd, yd, c = shape
    dev = torch.device("cuda")
    for dt in (torch.float32, torch.float64):
        x = torch.empty((0, c), dtype=dt, device=dev)
        w = torch.rand(xd * yd, c, dtype=torch.float64, device=dev)
        w0 = w.clone()
        BatchSOMTrainer(xd, yd, c, dev, batch_steps=8).train(x, w, num_passes=1)
        torch.cuda.synchronize()
        assert torch.equal(w, w0)
        labels, _ = som_device.assign(x, w)
        assert labels.numel() == 0
        sums, counts = som_device.cluster_sums(x, labels, xd * yd)
        assert float(sums.abs().sum()) == 0.0 and int(counts.sum()) == 0
        few = torch.rand((3, c), dtype=torch.float64, device=dev).to(dt)
        BatchSOMTrainer(xd, yd, c, dev, batch_steps=8).train(few, w, num_passes=1)
        torch.cuda.synchronize()
        assert torch.isfinite(w).all()
