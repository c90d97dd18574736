"""Mini-batch schedules of the batch rule (ark_analysis_amd/schedule.py; oracle orc_som_batch_sched): the host-side
arithmetic every backend shares, the oracle's two entry points against each other, and a two-rank gloo run on the
default two-phase schedule against the single-process oracle.  CPU only."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ark_analysis_amd.distributed import BatchSOMTrainer, broadcast_codebook
from ark_analysis_amd.flowsom import default_radius_range
from ark_analysis_amd.schedule import BatchSchedule, resolve
from tests.oracle_backend import OracleKernels
from tests.test_distributed_gloo import _free_port


def test_schedule_partitions_the_rows():
    for sch in (BatchSchedule.equal(7), BatchSchedule.two_phase(), BatchSchedule(12, [0, 5, 8, 9, 9, 12]),
                BatchSchedule.two_phase(head_steps=3, tail_steps=5, head_ratio=0.5, tail_phases_per_step=2)):
        for n in (0, 1, 11, 12, 13, 1000, 5003):
            parts = [sch.rows_of_step(n, g) for g in range(sch.steps)]
            allrows = np.concatenate(parts) if parts else np.empty(0, dtype=np.int64)
            assert np.array_equal(np.sort(allrows), np.arange(n)), (sch, n)
            for g, p in enumerate(parts):
                assert np.all(np.diff(p) > 0)
                assert np.all((p % sch.phases >= sch.edges[g]) & (p % sch.phases < sch.edges[g + 1]))


def test_default_schedule_shape_and_positions():
    sch = resolve(None)
    assert sch == BatchSchedule.two_phase() == resolve("two-phase") and sch.steps == 22 and sch.phases == 960
    widths = np.diff(sch.edges)
    assert np.all(widths[:6] >= widths[1:7]) and np.all(widths[6:-1] == widths[6]) and widths[-1] == 5 * widths[6]
    assert widths[:6].sum() * 6 == sch.phases * 5
    assert sch.position(0) == 0 and sch.position(6) * 6 == sch.phases * 5         # the tail starts where the radius reaches 1
    assert sch.position(sch.steps + 3) == sch.phases + sch.edges[3]               # second pass
    assert resolve(64) == BatchSchedule.equal(64) and resolve(64).position(70) == 70
    for bad in (0, -3, "fast", 2.5, True):
        with pytest.raises(ValueError, match="batch_steps"):
            resolve(bad)
    with pytest.raises(ValueError):
        BatchSchedule(10, [0, 4, 3, 10])


def test_oracle_equal_steps_are_the_scheduled_rule(oracle):
    rs = np.random.RandomState(3)
    x = rs.gamma(0.7, 0.5, size=(3001, 6))
    w0 = x[rs.choice(3001, 25, replace=False)].copy()
    rr = default_radius_range(5, 5)
    a = oracle.som_batch(x, w0, 5, 5, 2, (0.05, 0.01), rr, 16)
    b = oracle.som_batch_sched(x, w0, 5, 5, 2, (0.05, 0.01), rr, 16, list(range(17)))
    assert np.array_equal(a, b)
    # a step that takes two phases is NOT two steps: the statistics of both share one codebook
    c = oracle.som_batch_sched(x, w0, 5, 5, 2, (0.05, 0.01), rr, 16, list(range(0, 17, 2)))
    assert not np.array_equal(a, c)


def _worker(rank, world, port, shards, w0, xdim, ydim, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x = torch.from_numpy(shards[rank])
    w = torch.from_numpy(w0.copy()) if rank == 0 else torch.zeros(w0.shape, dtype=torch.float64)
    broadcast_codebook(w, 0)
    # (the two-phase schedule named explicitly: on a table this small the DEFAULT would resolve to equal steps)
    BatchSOMTrainer(xdim, ydim, x.shape[1], "cpu", batch_steps=BatchSchedule.two_phase(), kernels=OracleKernels()).train(x, w, num_passes=1)
    gathered = [torch.zeros_like(w) for _ in range(world)]
    dist.all_gather(gathered, w)
    if rank == 0:
        np.savez(out_path, w=w.numpy(), same=np.array([bool(torch.equal(g, w)) for g in gathered]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_the_default_schedule_match_the_oracle(oracle, tmp_path):
    """Each rank deals ITS rows into the schedule's phases; the union of the ranks' steps is the single-process
    step over the rows interleaved in blocks of `phases`."""
    xdim = ydim = 5
    sch = BatchSchedule.two_phase()
    k, c, n_local = 25, 5, 2 * sch.phases
    rs = np.random.RandomState(1)
    # values on a 2^-10 grid: every partial sum is exact in binary64, so the order the ranks' statistics are added in
    # cannot move a near-tie of the degenerate first steps (all nodes close to the global mean) to the other side
    shards = [rs.randint(0, 2048, size=(n_local, c)) / 1024.0 for _ in range(2)]
    w0 = shards[0][rs.choice(n_local, k, replace=False)].copy()
    out = str(tmp_path / "rank0.npz")
    mp.spawn(_worker, args=(2, _free_port(), shards, w0, xdim, ydim, out), nprocs=2, join=True)
    res = np.load(out)
    assert res["same"].all(), "codebook differs between ranks"
    blocks = [shards[r][j * sch.phases:(j + 1) * sch.phases] for j in range(n_local // sch.phases) for r in range(2)]
    want = oracle.som_batch_sched(np.concatenate(blocks), w0, xdim, ydim, 1, (0.05, 0.01), default_radius_range(xdim, ydim),
                                  sch.phases, sch.edges)
    np.testing.assert_allclose(res["w"], want, rtol=1e-10, atol=0)


def test_default_schedule_on_a_small_table_is_equal_steps(oracle):
    """train_mode="batch" with no schedule named: the two-phase default (960 phases) from 7 680 rows on, equal steps --
    at most 64, a row per node and step where the table allows -- below (a few thousand cells: the tail steps of the
    two-phase schedule would hold next to nothing, below 960 rows nothing at all)."""
    rs = np.random.RandomState(5)
    for n, steps in [(500, 20), (3_000, 64), (30, 1)]:
        x = rs.randint(0, 2048, size=(n, 4)) / 1024.0
        w0 = x[rs.choice(n, 25, replace=False)].copy()
        tr = BatchSOMTrainer(5, 5, 4, "cpu", kernels=OracleKernels())
        assert tr.schedule == BatchSchedule.two_phase()
        w = torch.from_numpy(w0.copy())
        tr.train(torch.from_numpy(x), w, num_passes=1)
        assert tr.schedule == BatchSchedule.equal(steps) and tr.batch_steps == steps
        want = oracle.som_batch(x, w0, 5, 5, 1, (0.05, 0.01), default_radius_range(5, 5), steps)
        np.testing.assert_allclose(w.numpy(), want, rtol=1e-12, atol=0)
    big = rs.randint(0, 2048, size=(8_000, 4)) / 1024.0
    tr = BatchSOMTrainer(5, 5, 4, "cpu", kernels=OracleKernels())
    tr.train(torch.from_numpy(big), torch.from_numpy(big[:25].copy()), num_passes=1)
    assert tr.schedule == BatchSchedule.two_phase()
    explicit = BatchSOMTrainer(5, 5, 4, "cpu", batch_steps=BatchSchedule.two_phase(), kernels=OracleKernels())
    explicit.train(torch.from_numpy(big[:500]), torch.from_numpy(big[:25].copy()), num_passes=1)
    assert explicit.schedule == BatchSchedule.two_phase()        # a schedule the caller named is the caller's


def _failing_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ark_analysis_amd import distributed, flowsom
    from tests import oracle_backend
    oracle_backend.install(setattr)
    outcome = []
    # (1) work only rank 0 does in front of a collective fails: every rank gets the exception, nobody waits in the broadcast
    try:
        distributed.on_rank0(lambda: (_ for _ in ()).throw(ValueError("rank 0 could not train")))
    except ValueError as e:
        outcome.append(str(e))
    # (2) som_batch on a job with fewer rows than nodes: decided on the JOB's row count, the same error on every rank
    rs = np.random.RandomState(rank)
    try:
        flowsom.som_batch(rs.rand(3, 4), xdim=3, ydim=3, seed=1, batch_steps=2)
    except ValueError as e:
        outcome.append(str(e))
    # (3) rank 0's share is shorter than K but the job has enough rows: initial nodes come from the pooled candidates
    n_local = 4 if rank == 0 else 40
    w = flowsom.som_batch(rs.rand(n_local, 4), xdim=3, ydim=3, seed=1, batch_steps=2)
    outcome.append(w.shape)
    gathered = [None] * world
    dist.all_gather_object(gathered, w.tobytes())
    outcome.append(len(set(gathered)) == 1)
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), np.array(outcome, dtype=object), allow_pickle=True)
    dist.barrier()
    dist.destroy_process_group()


def test_rank0_failures_reach_every_rank_and_short_shards_still_train(tmp_path):
    """Round-2 advice: rank-0-only work in front of a collective must not strand the other ranks; `n >= K` is a property of
    the job, not of rank 0's shard."""
    mp.spawn(_failing_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for rank in range(2):
        got = np.load(str(tmp_path / ("rank%d.npy" % rank)), allow_pickle=True)
        assert got[0] == "rank 0 could not train"
        assert "at least as many rows (6) as nodes (9)" in got[1]
        assert tuple(got[2]) == (9, 4) and bool(got[3])
