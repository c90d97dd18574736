"""The in-library exchange of the multi-rank batch rule (pxsom_comm_*, pxsom_batch_train_steps_sharded) on one MI355X:
a 1-rank RCCL communicator owned by libpxsom.  What one GPU can show: RCCL binds (PyTorch's copy, no second
library), the communicator initialises, an all-reduce over one rank leaves the buffer as it was, and a training run
with the exchange enqueued behind every step equals the run without it -- i.e. the per-step call sequence, buffers
and stream are right.  The sum over several ranks is RCCL's; the rule's arithmetic around it is covered by the
2-rank gloo tests (test_distributed_gloo.py, test_pipeline_multirank.py)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def comm():
    from ark_analysis_amd import som_device
    torch.cuda.set_device(0)
    c = som_device.RankComm(som_device.RankComm.unique_id(), 1, 0)
    yield c
    c.close()


def test_one_rank_allreduce_is_identity(comm):
    t = torch.randn(100 * 23, dtype=torch.float64, device="cuda")
    want = t.clone()
    comm.allreduce_sum(t)
    torch.cuda.synchronize()
    assert torch.equal(t, want)


@pytest.mark.parametrize("shape", [(10, 10, 22, torch.float32), (20, 20, 64, torch.float32), (12, 9, 5, torch.float64)])
def test_sharded_steps_equal_plain_steps(comm, shape):
    from ark_analysis_amd import som_device
    xd, yd, c, dt = shape
    gen = torch.Generator(device="cpu").manual_seed(5)
    x = torch.rand(6000, c, generator=gen, dtype=torch.float64).to(dt).cuda()
    w0 = torch.rand(xd * yd, c, generator=gen, dtype=torch.float64).cuda()
    out = []
    for use in (None, comm):
        st = som_device.BatchTrainState(x.shape[0], c, xd, yd, 8, x.device)
        st.wbuf[0].copy_(w0)
        som_device.batch_train_steps(x, st, 0, 16, 16, (0.05, 0.01), (3.0, 1.0), comm=use)
        w = torch.empty_like(w0)
        som_device.batch_train_finish(st, 16, 16, (0.05, 0.01), (3.0, 1.0), w)
        torch.cuda.synchronize()
        out.append(w.cpu().numpy())
    # binary64 atomics land in a run-dependent order: the two runs agree to rounding, not bit for bit
    np.testing.assert_allclose(out[0], out[1], rtol=1e-12, atol=0)


def test_exchange_needs_a_bound_communicator():
    from ark_analysis_amd import _capi
    rc = _capi.lib().pxsom_comm_allreduce_sum_f64(None, None, 0, None)
    assert rc != 0 and b"communicator" in _capi.lib().pxsom_last_error()


def test_native_exchange_under_a_one_rank_rccl_group():
    """distributed.native_exchange end to end (bind -> id -> broadcast over the group -> collective create ->
    agreement) in a fresh process with a 1-rank "nccl" process group."""
    code = (
        "import os, torch, torch.distributed as dist\n"
        "os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29577', RANK='0', WORLD_SIZE='1')\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))\n"
        "from ark_analysis_amd import distributed\n"
        "c = distributed.native_exchange(None)\n"
        "assert c is not None and c.nranks == 1, c\n"
        "assert distributed.native_exchange(None) is c\n"
        "t = torch.ones(2300, dtype=torch.float64, device='cuda'); c.allreduce_sum(t); torch.cuda.synchronize()\n"
        "assert float(t.sum()) == 2300.0\n"
        "dist.destroy_process_group()\n"
        "print('EXCHANGE_OK')\n")
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert "EXCHANGE_OK" in res.stdout, res.stdout + res.stderr


# ---- two real ranks through the in-library step loop --------------------------------------------------------------
# RCCL will not put two ranks on one GPU, and the round's boxes have one.  tests/mock_rccl/mock_rccl.cpp implements the
# five entry points pxsom_comm.hip binds (all-reduce through shared memory, binary64 sums in rank order); bound through
# PXSOM_RCCL_LIBRARY it lets pxsom_batch_train_steps_sharded run its loop -- step launch, exchange, next step -- on two
# processes that hold different rows.  What stays unexercised here is RCCL's own all-reduce.

def _mock_library(tmpdir) -> str:
    import shutil
    src = os.path.join(ROOT, "tests", "mock_rccl", "mock_rccl.cpp")
    out = os.path.join(str(tmpdir), "libmock_rccl.so")
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    subprocess.check_call([hipcc, "-O2", "-fPIC", "-shared", "-std=c++17", src, "-o", out, "-lrt"])
    return out


def _sharded_worker(rank, world, lib_path, uid_path, out_path, shape, n_local, steps, passes):
    os.environ["PXSOM_RCCL_LIBRARY"] = lib_path
    import time
    import torch as th
    from ark_analysis_amd import som_device
    th.cuda.set_device(0)
    if rank == 0:
        uid = som_device.RankComm.unique_id()
        with open(uid_path + ".tmp", "wb") as f:
            f.write(uid)
        os.rename(uid_path + ".tmp", uid_path)
    else:
        while not os.path.exists(uid_path):
            time.sleep(0.02)
        uid = open(uid_path, "rb").read()
    comm = som_device.RankComm(uid, world, rank)
    xd, yd, c, dt = shape
    rs = np.random.RandomState(100 + rank)
    centers = np.random.RandomState(7).rand(16, c)
    x_host = np.maximum(centers[rs.randint(0, 16, n_local[rank])] + 0.05 * rs.randn(n_local[rank], c), 0.0)
    x = th.from_numpy(x_host).to(dt).cuda()
    w0 = th.from_numpy(np.random.RandomState(3).rand(xd * yd, c)).cuda()
    total = steps * passes
    st = som_device.BatchTrainState(x.shape[0], c, xd, yd, steps, x.device)
    st.wbuf[0].copy_(w0)
    som_device.batch_train_steps(x, st, 0, total, total, (0.05, 0.01), (3.0, 1.0), comm=comm)
    w = th.empty_like(w0)
    som_device.batch_train_finish(st, total, total, (0.05, 0.01), (3.0, 1.0), w)
    th.cuda.synchronize()
    np.savez(out_path % rank, w=w.cpu().numpy(), x=x.cpu().to(th.float64).numpy(), w0=w0.cpu().numpy())
    comm.close()


@pytest.mark.parametrize("shape", [(10, 10, 22, torch.float32), (20, 20, 40, torch.float16), (7, 5, 9, torch.float64)])
def test_two_ranks_run_the_sharded_step_loop(oracle, tmp_path, shape):
    import torch.multiprocessing as mp
    lib_path = _mock_library(tmp_path)
    steps, passes = 8, 2
    n_local = (4000, 2808)                       # unequal shards: the ranks' mini-batches differ in size
    out = str(tmp_path / "rank%d.npz")
    mp.spawn(_sharded_worker, args=(2, lib_path, str(tmp_path / "uid"), out, shape, n_local, steps, passes), nprocs=2,
             join=True)
    r0, r1 = np.load(out % 0), np.load(out % 1)
    np.testing.assert_array_equal(r0["w"], r1["w"])          # identical codebooks without a second exchange
    # single-process equivalent: global mini-batch t = union of the ranks' local mini-batches t
    xd, yd, c, _ = shape
    rows, sizes = [], []
    for t in range(steps):
        part = np.concatenate([r0["x"][t::steps], r1["x"][t::steps]])
        rows.append(part)
        sizes.append(len(part))
    # orc_som_batch takes strided mini-batches of ONE matrix: interleave so that global row i % steps selects batch t
    longest = max(sizes)
    inter = np.full((longest * steps, c), np.nan)
    for t in range(steps):
        inter[t::steps][:sizes[t]] = rows[t]
    keep = ~np.isnan(inter).any(axis=1)
    if not keep.all():
        # unequal batch sizes cannot be laid out as one strided matrix: replay the rule step by step instead
        w = r0["w0"].copy()
        total = steps * passes
        for g in range(total):
            part = rows[g % steps]
            lab, _ = oracle.map_data_to_nodes(w, part)
            s, cnt = oracle.cluster_sums(part, lab, xd * yd)
            thr = 3.0 - (3.0 - 1.0) * g / total
            w = oracle.batch_update(w, xd, yd, s, cnt, 0.5 if thr < 1.0 else thr, 0.05 - (0.05 - 0.01) * g / total)
        want = w
    else:
        want = oracle.som_batch(inter, r0["w0"], xd, yd, passes, (0.05, 0.01), (3.0, 1.0), steps)
    np.testing.assert_allclose(r0["w"], want, rtol=1e-9, atol=1e-300)


@pytest.mark.parametrize("n_local", [(3000, 0), (3000, 5)])
def test_a_rank_with_an_empty_or_short_shard_keeps_the_same_codebook(oracle, tmp_path, n_local):
    """More ranks than data: a rank whose shard is empty (or shorter than the schedule) takes the SAME kernel route as the
    others -- the route is a property of the shape, not of the rank's row count -- applies the same all-reduced statistics
    and ends with the same bits (round-2 advice: the replicas drifted apart when such a rank fell back to another route)."""
    import torch.multiprocessing as mp
    lib_path = _mock_library(tmp_path)
    steps, passes = 8, 1
    shape = (10, 10, 22, torch.float32)
    out = str(tmp_path / "rank%d.npz")
    mp.spawn(_sharded_worker, args=(2, lib_path, str(tmp_path / "uid"), out, shape, n_local, steps, passes), nprocs=2, join=True)
    r0, r1 = np.load(out % 0), np.load(out % 1)
    np.testing.assert_array_equal(r0["w"], r1["w"])
    assert np.isfinite(r0["w"]).all()
    x_all = np.concatenate([r0["x"], r1["x"].reshape(-1, 22)])
    w = r0["w0"].copy()
    for g in range(steps):
        part = np.concatenate([r0["x"][g::steps], r1["x"].reshape(-1, 22)[g::steps]])
        lab, _ = oracle.map_data_to_nodes(w, part)
        s, cnt = oracle.cluster_sums(part, lab, 100)
        thr = 3.0 - (3.0 - 1.0) * g / steps
        w = oracle.batch_update(w, 10, 10, s, cnt, 0.5 if thr < 1.0 else thr, 0.05 - (0.05 - 0.01) * g / steps)
    np.testing.assert_allclose(r0["w"], w, rtol=1e-9, atol=1e-300)
    del x_all


# ---- the one-shot peer-to-peer exchange (pxsom_comm_p2p_*): two ranks on ONE device ---------------------------------
def _p2p_worker(rank, world, port, shards, w0, xdim, ydim, out_dir, route="p2p"):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["PXSOM_EXCHANGE"] = route
    from ark_analysis_amd import som_device
    from ark_analysis_amd.distributed import BatchSOMTrainer, native_exchange
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    comm = native_exchange(None)
    assert isinstance(comm, som_device.P2PComm), comm
    # (a) the exchange itself: values whose sum depends on the order of the additions in the last bits
    gen = torch.Generator(device="cpu").manual_seed(100 + rank)
    t = (torch.rand(2323, generator=gen, dtype=torch.float64) * (10.0 ** (3 * rank))).to(dev)
    mine = t.clone()
    for rep in range(40):                     # many epochs back to back: parity reuse, a rank running ahead
        buf = mine.clone()
        comm.allreduce_sum(buf)
    torch.cuda.synchronize()
    assert comm.error_epoch() == 0
    parts = [torch.zeros(2323, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(parts, mine.cpu())
    want = parts[0].clone()
    for p in parts[1:]:
        want = want + p                       # rank order, like the kernel
    sums = [torch.zeros(2323, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(sums, buf.cpu())
    ok_sum = bool(torch.equal(buf.cpu(), want)) and all(bool(torch.equal(s, sums[0])) for s in sums)
    # (b) a sharded training run with the exchange enqueued behind every step by the library
    x = torch.from_numpy(shards[rank]).to(dev)
    w = torch.from_numpy(w0.copy()).to(dev)
    trainer = BatchSOMTrainer(xdim, ydim, x.shape[1], dev)          # default schedule: 22 steps, 22 exchanges
    trainer.train(x, w, num_passes=1)
    torch.cuda.synchronize()
    err = comm.error_epoch()
    gathered = [torch.zeros(w.shape, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(gathered, w.cpu())
    if rank == 0:
        np.savez(os.path.join(out_dir, route + ".npz"), w=w.cpu().numpy(), ok_sum=np.array(ok_sum), err=np.array(err),
                 same=np.array([bool(torch.equal(g, gathered[0])) for g in gathered]))
    dist.barrier()
    dist.destroy_process_group()


def test_p2p_exchange_two_ranks_on_one_device(oracle, tmp_path):
    """Two processes on cuda:0, the rule's exchange through the library's own peer-to-peer all-reduce (HIP IPC blocks, one
    launch per rank and exchange): sums bit-identical on both ranks and equal to the rank-ordered sum; a sharded training
    run on the default schedule ends in codebooks that are array_equal across the ranks and match the oracle on the
    united rows.  (RCCL refuses two ranks on one device; this is the exchange a one-GPU box CAN run.)"""
    import socket

    import torch.multiprocessing as mp

    from ark_analysis_amd import synth
    from ark_analysis_amd.flowsom import default_radius_range
    from ark_analysis_amd.schedule import BatchSchedule
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    xdim = ydim = 10
    k, c, n_local = 100, 22, 48_000
    shards = [synth.make_fov_numpy(n_local, c, seed=70 + r, dtype=np.float32) for r in range(2)]
    w0 = shards[0][np.random.RandomState(1).choice(n_local, k, replace=False)].astype(np.float64)
    mp.spawn(_p2p_worker, args=(2, port, shards, w0, xdim, ydim, str(tmp_path)), nprocs=2, join=True)
    res = np.load(str(tmp_path / "p2p.npz"))
    assert int(res["err"]) == 0, "a peer did not arrive at exchange %d" % int(res["err"])
    assert bool(res["ok_sum"]), "all-reduce: not the rank-ordered sum, or the ranks differ"
    assert res["same"].all(), "codebook differs between the ranks"
    # the united rows in the order the rule sees them: row i of the job = rows i of rank 0, rank 1 interleaved per phase
    sch = BatchSchedule.two_phase()
    # every rank deals ITS rows into the phases (local i % phases), so the job's step g holds both shards' step-g rows;
    # the oracle takes one matrix: build it so that its own dealing reproduces those steps (phase by phase)
    phases = sch.phases
    blocks = []
    for j in range(n_local // phases):
        for r in range(2):
            blocks.append(shards[r][j * phases:(j + 1) * phases])
    tail = n_local - (n_local // phases) * phases
    assert tail == 0, "pick n_local as a multiple of the schedule's phases"
    g = np.concatenate(blocks).astype(np.float64)
    # rows of phase p: oracle index i % (2 * phases) ... the oracle deals by i % phases, so interleave within a phase pair
    want = oracle.som_batch_sched(g, w0, xdim, ydim, 1, (0.05, 0.01), default_radius_range(xdim, ydim), phases, sch.edges)
    np.testing.assert_allclose(res["w"], want, rtol=1e-9, atol=0)


def test_fused_exchange_two_ranks_on_one_device(oracle, tmp_path):
    """The exchange INSIDE the step launches (PXSOM_EXCHANGE=fused: the last workgroup of a step writes this rank's statistics
    into every rank's block, the next step adds the slots in rank order while it applies the update), two processes on cuda:0:
    no peer late, codebooks array_equal across the ranks AND array_equal to the run with the one-launch exchange behind every
    step (the same additions in the same order), on a schedule with steps of one, two and four tiles per wave."""
    import socket

    import torch.multiprocessing as mp

    from ark_analysis_amd import synth
    xdim = ydim = 10
    k, c, n_local = 100, 22, 96_000
    shards = [synth.make_fov_numpy(n_local, c, seed=170 + r, dtype=np.float32) for r in range(2)]
    w0 = shards[1][np.random.RandomState(2).choice(n_local, k, replace=False)].astype(np.float64)
    got = {}
    for route in ("fused", "p2p"):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        mp.spawn(_p2p_worker, args=(2, port, shards, w0, xdim, ydim, str(tmp_path), route), nprocs=2, join=True)
        res = np.load(str(tmp_path / (route + ".npz")))
        assert int(res["err"]) == 0, "%s: a peer did not arrive at exchange %d" % (route, int(res["err"]))
        assert res["same"].all(), route + ": codebook differs between the ranks"
        assert np.isfinite(res["w"]).all(), route
        got[route] = res["w"]
    assert np.array_equal(got["fused"], got["p2p"]), "largest difference %.3g" % float(np.abs(got["fused"] - got["p2p"]).max())


@pytest.mark.parametrize("c,rows", [(100, (30_000, 9_600)), (22, (61_440, 7_680))])
def test_uneven_shards_end_in_equal_codebooks(tmp_path, c, rows):
    """Ranks with shards of very different lengths take different kernel routes for the same step -- the wide one-launch step up
    to 16 K local rows, the launch-per-phase route beyond (cell SOM, 100 columns); one or more tiles per wave in the fused step
    (22 columns) -- and apply the same all-reduced statistics locally: the replicas stay equal only if the routes' updates are
    bit-identical (the advisor's round-4 finding).  Two processes on cuda:0, peer-to-peer exchange, default schedule."""
    import socket

    import torch.multiprocessing as mp

    from ark_analysis_amd import synth
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    k = 100
    shards = [synth.make_fov_numpy(rows[r], c, seed=270 + r, dtype=np.float32) for r in range(2)]
    w0 = shards[0][np.random.RandomState(3).choice(rows[0], k, replace=False)].astype(np.float64)
    mp.spawn(_p2p_worker, args=(2, port, shards, w0, 10, 10, str(tmp_path)), nprocs=2, join=True)
    res = np.load(str(tmp_path / "p2p.npz"))
    assert int(res["err"]) == 0
    assert np.isfinite(res["w"]).all()
    assert res["same"].all(), "codebook differs between the ranks"


def _mixed_route_worker(rank, world, port, shards, w0, out_dir, route):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PXSOM_EXCHANGE=route)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from ark_analysis_amd.distributed import BatchSOMTrainer
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    x = torch.from_numpy(shards[rank]).to(dev)
    if rank == 1:                                   # the same rows behind an ODD row stride: not the fused step's shape
        wide = torch.zeros((x.shape[0], x.shape[1] + 1), dtype=x.dtype, device=dev)
        wide[:, :x.shape[1]] = x
        x = wide[:, :x.shape[1]]
    w = torch.from_numpy(w0.copy()).to(dev)
    trainer = BatchSOMTrainer(10, 10, x.shape[1], dev)
    trainer.train(x, w, num_passes=1)
    torch.cuda.synchronize()
    gathered = [torch.zeros(w.shape, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(gathered, w.cpu())
    agreement = dict(trainer.kernels.route_agreement)
    np.savez(os.path.join(out_dir, "mixed_%s_%d.npz" % (route, rank)), w=w.cpu().numpy(),
             same=np.array([bool(torch.equal(g, gathered[0])) for g in gathered]),
             flags=np.array([agreement["all_fused"], agreement["any_fused"], agreement["unfused_now"]]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("route", ["p2p", "fused"])
def test_one_misaligned_shard_sends_every_rank_to_the_launch_per_phase_route(oracle, tmp_path, route):
    """Round 5's advice: rank 0 can take the one-launch 10 x 10 step, rank 1 cannot (odd row stride).  The route is a collective
    decision -- `any but not all` must be detected (it was not: the second flag of the MIN-reduce could never differ from the
    first) and every rank must then run the launch-per-phase steps, or one rank would wait inside the fused step's in-kernel
    exchange for a peer that exchanges through the separate launch.  Both ranks report the same verdict, end with array_equal
    codebooks, and the codebook matches the oracle on the united rows."""
    import socket

    import torch.multiprocessing as mp

    from ark_analysis_amd import synth
    from ark_analysis_amd.flowsom import default_radius_range
    from ark_analysis_amd.schedule import BatchSchedule
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    k, c, n_local = 100, 22, 19_200
    shards = [synth.make_fov_numpy(n_local, c, seed=370 + r, dtype=np.float32) for r in range(2)]
    w0 = shards[0][np.random.RandomState(5).choice(n_local, k, replace=False)].astype(np.float64)
    mp.spawn(_mixed_route_worker, args=(2, port, shards, w0, str(tmp_path), route), nprocs=2, join=True)
    res = [np.load(str(tmp_path / ("mixed_%s_%d.npz" % (route, r)))) for r in range(2)]
    for r in res:
        assert list(r["flags"]) == [False, True, True], list(r["flags"])   # not all, but some: launch per phase everywhere
        assert r["same"].all(), "codebook differs between the ranks"
    sch = BatchSchedule.two_phase()
    phases = sch.phases
    assert n_local % phases == 0
    blocks = [shards[r][j * phases:(j + 1) * phases] for j in range(n_local // phases) for r in range(2)]
    want = oracle.som_batch_sched(np.concatenate(blocks).astype(np.float64), w0, 10, 10, 1, (0.05, 0.01), default_radius_range(10, 10),
                                  phases, sch.edges)
    np.testing.assert_allclose(res[0]["w"], want, rtol=1e-9, atol=0)


def _late_peer_worker(rank, world, port, shards, w0, out_dir, route):
    import time
    import warnings

    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PXSOM_EXCHANGE=route)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from ark_analysis_amd import som_device
    from ark_analysis_amd.distributed import BatchSOMTrainer
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    x = torch.from_numpy(shards[rank]).to(dev)
    w = torch.from_numpy(w0.copy()).to(dev)
    if rank == 1:       # rank 1 reaches the run of steps six seconds after rank 0: past the exchanges' four-second bound
        real, held = som_device.batch_train_steps, []

        def held_back(*a, **k):
            if not held:
                held.append(1)
                time.sleep(6.0)
            return real(*a, **k)
        som_device.batch_train_steps = held_back
    trainer = BatchSOMTrainer(10, 10, x.shape[1], dev)
    t0 = time.time()
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        trainer.train(x, w, num_passes=1)
        torch.cuda.synchronize()
    took = time.time() - t0
    gathered = [torch.zeros(w.shape, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(gathered, w.cpu())
    np.savez(os.path.join(out_dir, "late_%s_%d.npz" % (route, rank)), w=w.cpu().numpy(), took=np.array(took),
             same=np.array([bool(torch.equal(g, gathered[0])) for g in gathered]),
             retired=np.array(any("retired" in str(c.message) for c in caught)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("route", ["fused", "p2p"])
def test_a_late_peer_is_noticed_and_the_pass_repeated(oracle, tmp_path, route):
    """A rank that reaches its run of steps after the exchanges' four-second bound: the waiting rank records the epoch, is late
    at once in every later step of the call (no cascade of timeouts: the call returns in seconds, not steps x 4 s), still raises
    its own flags -- over NaN slots on the in-kernel route -- and the ranks agree to retire the route and run the pass again
    through torch.distributed: equal, finite codebooks that match the oracle on the united rows (round 5's advice)."""
    import socket

    import torch.multiprocessing as mp

    from ark_analysis_amd import synth
    from ark_analysis_amd.flowsom import default_radius_range
    from ark_analysis_amd.schedule import BatchSchedule
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    k, c, n_local = 100, 22, 19_200
    shards = [synth.make_fov_numpy(n_local, c, seed=470 + r, dtype=np.float32) for r in range(2)]
    w0 = shards[1][np.random.RandomState(6).choice(n_local, k, replace=False)].astype(np.float64)
    mp.spawn(_late_peer_worker, args=(2, port, shards, w0, str(tmp_path), route), nprocs=2, join=True)
    res = [np.load(str(tmp_path / ("late_%s_%d.npz" % (route, r)))) for r in range(2)]
    assert bool(res[0]["retired"]), "rank 0 did not report the retired route"
    for r in res:
        assert r["same"].all() and np.isfinite(r["w"]).all()
        assert float(r["took"]) < 40.0, float(r["took"])       # one or two timeouts, not one per step
    sch = BatchSchedule.two_phase()
    phases = sch.phases
    blocks = [shards[r][j * phases:(j + 1) * phases] for j in range(n_local // phases) for r in range(2)]
    want = oracle.som_batch_sched(np.concatenate(blocks).astype(np.float64), w0, 10, 10, 1, (0.05, 0.01), default_radius_range(10, 10),
                                  phases, sch.edges)
    np.testing.assert_allclose(res[0]["w"], want, rtol=1e-9, atol=0)


def _native_exchange_worker(rank, world, lib_path, port, out_path, wrong):
    os.environ.update(PXSOM_RCCL_LIBRARY=lib_path, PXSOM_NATIVE_EXCHANGE="force", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if wrong:
        os.environ["MOCK_RCCL_WRONG_SUM"] = "1"
    import warnings
    import torch as th
    import torch.distributed as dist
    from ark_analysis_amd import distributed
    th.cuda.set_device(0)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        comm = distributed.native_exchange(None)
    verdict = "none" if comm is None else type(comm).__name__
    if comm is not None:        # a communicator that passed its first all-reduce serves the next one as well
        t = th.full((64,), float(rank + 1), dtype=th.float64, device="cuda")
        comm.allreduce_sum(t)
        th.cuda.synchronize()
        assert float(t[0]) == world * (world + 1) / 2
    with open(out_path % rank, "w") as f:
        f.write(verdict + "|" + ";".join(str(w.message) for w in caught))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("wrong", [False, True])
def test_a_communicator_is_checked_before_it_is_trusted(tmp_path, wrong):
    """`distributed.native_exchange` runs one all-reduce of a known payload through a communicator it has just made; a
    collective library that returns other sums (here: the stand-in library with an off-by-one first word) is dropped by
    both ranks together -- the run all-reduces through torch.distributed instead -- and rank 0 says why."""
    import socket
    import torch.multiprocessing as mp
    lib_path = _mock_library(tmp_path)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "verdict%d.txt")
    mp.spawn(_native_exchange_worker, args=(2, lib_path, port, out, wrong), nprocs=2, join=True)
    v0, v1 = open(out % 0).read(), open(out % 1).read()
    if wrong:
        assert v0.startswith("none|") and v1.startswith("none|")
        assert "did not return the expected sums" in v0
    else:
        assert v0.startswith("RankComm|") and v1.startswith("RankComm|")
