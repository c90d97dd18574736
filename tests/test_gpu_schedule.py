"""The scheduled batch rule on the HIP path (pxsom_batch_train_sched) against orc_som_batch_sched: per step (statistics
of the rows the schedule gives the step, for the codebook the step searched with; W_{g+1} against orc_batch_update),
fused route against launch-per-phase route, whole runs on the default two-phase schedule, and BASELINE config 3's
per-GPU share (25 FOVs x 1024^2 x 22) at full size.  `-m gpu` only."""
import numpy as np
import pytest
import torch

from ark_analysis_amd import som_device as sd
from ark_analysis_amd import synth
from ark_analysis_amd.distributed import BatchSOMTrainer, batch_schedule
from ark_analysis_amd.flowsom import default_radius_range
from ark_analysis_amd.schedule import BatchSchedule

pytestmark = pytest.mark.gpu

MIXED = BatchSchedule(12, [0, 5, 8, 9, 12])          # steps of 5, 3, 1, 3 phases
SMALL_TWO_PHASE = BatchSchedule.two_phase(head_steps=3, tail_steps=5, head_ratio=0.5, tail_phases_per_step=2)


def _codebook(x, k, seed):
    rs = np.random.RandomState(seed)
    return np.ascontiguousarray(x[rs.choice(x.shape[0], size=k, replace=False)].astype(np.float64))


@pytest.mark.parametrize("c,dtype,n,grid,sch,passes", [
    (22, np.float32, 40_003, 10, MIXED, 1),              # fused kernel, two-level row view, n % phases != 0
    (22, np.float32, 9_001, 10, SMALL_TWO_PHASE, 2),     # two passes
    (16, np.float16, 30_000, 10, MIXED, 1),
    (22, np.float64, 10_000, 10, MIXED, 1),              # the drop-in classes' dtype
    (8, np.float32, 7, 10, MIXED, 1),                    # fewer rows than phases: empty steps
    (100, np.float32, 30_001, 10, MIXED, 1),             # generic route: rows gathered into step order
    (40, np.float16, 40_000, 20, SMALL_TWO_PHASE, 1),
    (7, np.float32, 12_345, 10, MIXED, 1),               # odd channel count: 2-byte / 4-byte gather chunks
    (40, np.float64, 9_000, 20, MIXED, 2),
    (1, np.float32, 17_474, 16, SMALL_TWO_PHASE, 1),     # one channel (round 6: the wide step's e / c by a multiplication has no 2^32 / 1)
    (1, np.float64, 2_999, 10, MIXED, 1),
])
def test_scheduled_steps_match_the_oracle_per_step(gpu, oracle, c, dtype, n, grid, sch, passes):
    xdim = ydim = grid
    k = xdim * ydim
    x = synth.make_fov_numpy(max(n, 2 * k), c, seed=31, dtype=np.float32).astype(dtype)[:n]
    w0 = _codebook(synth.make_fov_numpy(4 * k, c, seed=32, dtype=np.float64), k, seed=5)
    w0[k - 3] = w0[1]                                   # a duplicate node
    xd = torch.from_numpy(x).to(gpu)
    x64 = x.astype(np.float64)
    rr = default_radius_range(xdim, ydim)
    total = passes * sch.steps
    exact_rows = dtype != np.float64                    # binary64 rows: atomic summation order leaves 1e-16 noise
    states = [sd.BatchTrainState(n, c, xdim, ydim, sch, gpu, dtype=xd.dtype) for _ in range(2)]
    for st in states:
        st.wbuf[0].copy_(torch.from_numpy(w0))
    w_prev = s_prev = cnt_prev = None
    for g in range(total):
        sd.batch_train_steps(xd, states[0], g, g + 1, total, (0.05, 0.01), rr)
        sd.batch_train_steps(xd, states[1], g, g + 1, total, (0.05, 0.01), rr, unfused=True)
        w_g = states[0].wbuf[g % 2].cpu().numpy()
        if g > 0:
            thr, alpha = batch_schedule(sch.position(g - 1), passes * sch.phases, (0.05, 0.01), rr)
            np.testing.assert_allclose(w_g, oracle.batch_update(w_prev, xdim, ydim, s_prev, cnt_prev, thr, alpha), rtol=1e-12, atol=0)
        rows = x64[sch.rows_of_step(n, g)]
        lab, _ = oracle.map_data_to_nodes(w_g, rows) if len(rows) else (np.empty(0, np.int32), None)
        s, cnt = oracle.cluster_sums(rows.reshape(-1, c), lab, k)
        ring = states[0].ring[g % 3].cpu().numpy()
        np.testing.assert_array_equal(ring[k * c:], cnt.astype(np.float64))
        if exact_rows:
            np.testing.assert_array_equal(ring[: k * c].reshape(k, c), s)
            assert torch.equal(states[0].wbuf[g % 2], states[1].wbuf[g % 2]), f"codebook of step {g}: routes differ"
            assert torch.equal(states[0].ring[g % 3], states[1].ring[g % 3]), f"statistics of step {g}: routes differ"
        else:
            np.testing.assert_allclose(ring[: k * c].reshape(k, c), s, rtol=1e-12, atol=1e-12)
            states[1].wbuf.copy_(states[0].wbuf)        # keep the two routes on one trajectory
            np.testing.assert_allclose(states[1].ring[g % 3].cpu().numpy(), ring, rtol=1e-12, atol=1e-12)
            states[1].ring.copy_(states[0].ring)
        assert float(states[0].ring[(g + 1) % 3].abs().max()) == 0.0, "next statistics buffer not cleared"
        w_prev, s_prev, cnt_prev = w_g, ring[: k * c].reshape(k, c).copy(), cnt
    wa = torch.empty((k, c), dtype=torch.float64, device=gpu)
    sd.batch_train_finish(states[0], total, total, (0.05, 0.01), rr, wa)
    if not exact_rows:
        return
    # the whole run in one call == step by step == the oracle's run
    st = sd.BatchTrainState(n, c, xdim, ydim, sch, gpu, dtype=xd.dtype)
    st.wbuf[0].copy_(torch.from_numpy(w0))
    sd.batch_train_steps(xd, st, 0, total, total, (0.05, 0.01), rr)
    wc = torch.empty_like(wa)
    sd.batch_train_finish(st, total, total, (0.05, 0.01), rr, wc)
    assert torch.equal(wa, wc)
    want = oracle.som_batch_sched(x64, w0, xdim, ydim, passes, (0.05, 0.01), rr, sch.phases, sch.edges)
    np.testing.assert_allclose(wa.cpu().numpy(), want, rtol=1e-9, atol=0)


@pytest.mark.parametrize("c,dtype,n,grid", [(22, np.float32, 300_000, 10), (40, np.float16, 120_000, 20), (100, np.float32, 60_000, 10)])
def test_default_schedule_run_matches_the_oracle(gpu, oracle, c, dtype, n, grid):
    """BatchSOMTrainer on its default (two-phase, 22 steps, 960 phases) schedule, one call, against orc_som_batch_sched;
    a strided view (rows with padding) takes the same route and gives the same codebook."""
    xdim = ydim = grid
    k = xdim * ydim
    x = synth.make_fov_numpy(n, c, seed=41, dtype=np.float32).astype(dtype)
    w0 = _codebook(x, k, seed=6)
    xd = torch.from_numpy(x).to(gpu)
    tr = BatchSOMTrainer(xdim, ydim, c, gpu)
    assert tr.schedule == BatchSchedule.two_phase() and tr.batch_steps == 22
    w = torch.from_numpy(w0.copy()).to(gpu)
    tr.train(xd, w, num_passes=1)
    want = oracle.som_batch_sched(x.astype(np.float64), w0, xdim, ydim, 1, (0.05, 0.01), default_radius_range(xdim, ydim),
                                  tr.schedule.phases, tr.schedule.edges)
    np.testing.assert_allclose(w.cpu().numpy(), want, rtol=1e-9, atol=0)
    padded = torch.zeros((n, c + 2), dtype=xd.dtype, device=gpu)
    padded[:, :c] = xd
    w2 = torch.from_numpy(w0.copy()).to(gpu)
    BatchSOMTrainer(xdim, ydim, c, gpu).train(padded[:, :c], w2, num_passes=1)
    assert torch.equal(w, w2)


def test_config3_share_at_full_size(gpu, oracle):
    """BASELINE configs[2]'s per-GPU share -- 25 FOVs x 1024^2 x 22 fp32 (2.3 GB), 10 x 10 SOM: a batch pass over the
    10 % subset on the default schedule ends finite and accounts for every training row, labels of all 26 M rows
    against the oracle on a 200 k-row sample, idempotence, and the count / sum checksums of the mean table."""
    fovs, p, c, xdim, ydim = 25, 1024 * 1024, 22, 10, 10
    k = xdim * ydim
    n = fovs * p
    x = torch.empty((n, c), dtype=torch.float32, device=gpu)
    for f in range(fovs):
        x[f * p:(f + 1) * p] = synth.make_fov_torch(p, c, seed=3000 + f, device=gpu)
    sub = x[::10].contiguous()
    g = torch.Generator(device="cpu")
    g.manual_seed(9)
    w = sub[torch.randperm(sub.shape[0], generator=g)[:k].to(gpu)].double().contiguous()
    tr = BatchSOMTrainer(xdim, ydim, c, gpu)
    tr.train(sub, w, num_passes=1)
    assert bool(torch.isfinite(w).all())
    st = tr.kernels._state
    last = st.ring[(tr.batch_steps - 1) % 3]
    assert int(last[k * c:].sum().item()) == len(tr.schedule.rows_of_step(sub.shape[0], tr.batch_steps - 1))
    labels, _ = sd.assign(x, w)
    labels2, _ = sd.assign(x, w)
    assert torch.equal(labels, labels2)
    assert int(labels.min()) >= 1 and int(labels.max()) <= k
    assert sd.last_exact_rows(sd.assign.last_workspace) < n // 20
    idx = torch.randperm(n, device=gpu)[:200_000]
    want, _ = oracle.map_data_to_nodes(w.cpu().numpy(), x[idx].double().cpu().numpy())
    np.testing.assert_array_equal(labels[idx].cpu().numpy(), want)
    sums, cnt = sd.cluster_sums(x, labels, k)
    assert int(cnt.sum()) == n
    np.testing.assert_array_equal(cnt.cpu().numpy(), torch.bincount(labels.long() - 1, minlength=k).cpu().numpy())
    np.testing.assert_allclose(sums.sum(dim=0).cpu().numpy(), x.sum(dim=0, dtype=torch.float64).cpu().numpy(), rtol=1e-9)


@pytest.mark.parametrize("c,grid,n,sch,unfused", [(22, 10, 30_000, MIXED, False), (22, 10, 30_000, MIXED, True), (100, 10, 20_001, MIXED, False),
                                                  (40, 20, 25_000, SMALL_TWO_PHASE, False), (400, 10, 4_000, MIXED, False)])
def test_binary64_rows_train_reproducibly(gpu, oracle, c, grid, n, sch, unfused):
    """binary64 rows with the run's quantum (include/pxsom.h "Reproducible statistics"): every step's statistics are the
    EXACT sums of the quantised rows -- equal to the oracle's bit for bit whatever the order the workgroups delivered
    them in --, two runs end in bit-identical codebooks (the reference pins same-seed retraining:
    tests/phenotyping/cluster_helpers_test.py:323-332), and the trainer picks the quantum itself for float64 input.
    Data with many near-identical rows and a crowded initial codebook: what used to let two runs part ways."""
    xdim = ydim = grid
    k = xdim * ydim
    rs = np.random.RandomState(c + n)
    x = synth.make_fov_numpy(n, c, seed=51, dtype=np.float64)
    x[1::3] = x[0::3][: len(x[1::3])] * (1.0 + 1e-13 * rs.standard_normal((len(x[1::3]), 1)))     # near-duplicates
    w0 = np.ascontiguousarray(x[rs.choice(n, k, replace=False)])
    w0[k // 2:] = w0[: k - k // 2] * (1.0 + 1e-12)                                                # near-identical node pairs
    xd = torch.from_numpy(x).to(gpu)
    rr = default_radius_range(xdim, ydim)
    widest = int(np.diff(sch.edges).max())
    q = sd.exact_sum_quantum(float(np.abs(x).max()), (n // sch.phases + 1) * widest)
    assert q > 0 and np.log2(q) == np.round(np.log2(q))
    assert float(sd.absmax(xd).item()) == float(np.abs(x).max())
    total = sch.steps
    runs = []
    for rep in range(2):
        st = sd.BatchTrainState(n, c, xdim, ydim, sch, gpu, dtype=xd.dtype)
        st.quantum = q
        st.wbuf[0].copy_(torch.from_numpy(w0))
        w_prev = s_prev = cnt_prev = None
        for g in range(total):
            sd.batch_train_steps(xd, st, g, g + 1, total, (0.05, 0.01), rr, unfused=unfused)
            w_g = st.wbuf[g % 2].cpu().numpy()
            ring = st.ring[g % 3].cpu().numpy()
            if rep == 0:
                if g > 0:
                    thr, alpha = batch_schedule(sch.position(g - 1), sch.phases, (0.05, 0.01), rr)
                    np.testing.assert_allclose(w_g, oracle.batch_update(w_prev, xdim, ydim, s_prev, cnt_prev, thr, alpha), rtol=1e-12, atol=1e-300)
                rows = x[sch.rows_of_step(n, g)]
                lab, _ = oracle.map_data_to_nodes(w_g, rows)
                s, cnt = oracle.cluster_sums(oracle.quantize(rows, q), lab, k)
                np.testing.assert_array_equal(ring[k * c:], cnt.astype(np.float64))
                np.testing.assert_array_equal(ring[: k * c].reshape(k, c), s)          # exact sums: bit for bit
                w_prev, s_prev, cnt_prev = w_g, s, cnt
        wa = torch.empty((k, c), dtype=torch.float64, device=gpu)
        sd.batch_train_finish(st, total, total, (0.05, 0.01), rr, wa)
        runs.append(wa.cpu().numpy())
    assert np.array_equal(runs[0], runs[1])
    # the trainer on float64 input: quantum chosen by itself, whole run in one call, twice
    outs = []
    for rep in range(2):
        w = torch.from_numpy(w0.copy()).to(gpu)
        tr = BatchSOMTrainer(xdim, ydim, c, gpu, batch_steps=sch)
        tr.train(xd, w, num_passes=1)
        assert tr.kernels._state.quantum == q
        outs.append(w.cpu().numpy())
    assert np.array_equal(outs[0], outs[1])
    if not unfused:
        assert np.array_equal(outs[0], runs[0])


def test_default_schedule_quality_against_equal_steps_and_online(gpu, oracle):
    """What the two-phase schedule is for: the quality of 64 equal steps in 22 launches.  Mean quantisation error (distance
    to the BMU over all rows) of the codebooks three rules reach from the same initial nodes on a 400 k-row mixture: the
    default schedule within 1 % of 64 equal steps and of the ONLINE oracle (the reference's rule); 32 EQUAL steps are
    measurably worse than both (DESIGN.md K6b holds the six-seed study)."""
    n, c, xdim, ydim = 400_000, 22, 10, 10
    k = xdim * ydim
    x = np.concatenate([synth.make_fov_numpy(n // 4, c, seed=900 + i, dtype=np.float32) for i in range(4)])
    rs = np.random.RandomState(77)
    w0 = np.ascontiguousarray(x[rs.choice(n, k, replace=False)].astype(np.float64))
    rr = default_radius_range(xdim, ydim)
    xd = torch.from_numpy(x).to(gpu)

    def qe(w_host):
        _, d = sd.assign(xd, torch.from_numpy(np.ascontiguousarray(w_host)).to(gpu), want_dists=True)
        return float(d.mean().item())

    def batch(spec):
        w = torch.from_numpy(w0.copy()).to(gpu)
        BatchSOMTrainer(xdim, ydim, c, gpu, batch_steps=spec).train(xd, w, num_passes=1)
        return qe(w.cpu().numpy())
    order = rs.randint(0, n, size=n).astype(np.int64)
    q_online = qe(oracle.som_online(x.astype(np.float64), w0, xdim, ydim, 1, (0.05, 0.01), rr, order))
    q_default, q_64, q_32 = batch(None), batch(64), batch(32)
    assert q_default <= 1.01 * q_64, (q_default, q_64)
    assert q_default <= 1.015 * q_online, (q_default, q_online)
    assert q_32 >= q_default, (q_32, q_default)


TAIL_HEAVY = BatchSchedule.two_phase(head_steps=2, tail_steps=9, head_ratio=0.5, tail_phases_per_step=3)


@pytest.mark.parametrize("c,dtype,n,sch,passes", [
    (22, np.float32, 120_007, None, 1),            # the default schedule: 22 one-launch steps
    (22, np.float32, 20_011, TAIL_HEAVY, 1),
    (16, np.float16, 31_000, SMALL_TWO_PHASE, 2),  # two passes: the tail is the end of the second
    (32, np.float64, 12_000, TAIL_HEAVY, 1),       # binary64 rows (rounded to the run's quantum), widest fused rows
    # a few rows per member and step, ~700 rows in a window: the gain 1 - (1 - alpha)^den sits within an ulp of 1.  Until
    # round 5 the gain went through expm1 (the device library's and glibc's round it to different sides here, and the crowded
    # first codebooks turn that bit into different BMUs); it is a chain of plain products now (batch_gain), the same bits on
    # both sides, and the case is compared with the oracle like the others
    (8, np.float32, 2_500, SMALL_TWO_PHASE, 1),
    (2, np.float32, 4_001, TAIL_HEAVY, 1),
])
def test_one_launch_steps_equal_the_launch_per_phase_route_and_the_oracle(gpu, oracle, c, dtype, n, sch, passes):
    """(Rounds 4 - 5: test_persistent_tail_equals_the_launch_per_step_route -- the opt-in persistent tail kernel it pinned was
    removed in round 6; its six cases stay.)  The one-launch fused steps against the launch-per-phase route
    (PXSOM_TRAIN_UNFUSED: update, prepare, search, exact and sums kernels per step): the codebook and the state left behind (W
    of the last step, its statistics, the cleared next buffer) bit for bit on data whose sums are exact, and the run against
    orc_som_batch_sched."""
    xdim = ydim = 10
    k = 100
    sch = BatchSchedule.two_phase() if sch is None else sch
    # values on a 2^-12 grid: every partial sum is exact, so the trajectory does not depend on the order the rows are
    # added in (with arbitrary binary32 values the collapsed codebooks of the first steps amplify a last-bit difference
    # between two summation orders into different BMUs, and an independent oracle run parts ways with any GPU run)
    x = synth.make_fov_numpy(max(n, 2 * k), c, seed=51, dtype=np.float32)[:n]
    x = (np.round(x.astype(np.float64) * 4096.0) / 4096.0).astype(dtype)
    w0 = _codebook(synth.make_fov_numpy(4 * k, c, seed=52, dtype=np.float64), k, seed=7)
    w0[k - 2] = w0[3]                                    # a duplicate node: not masked out by the BMU-only steps' filter
    xd = torch.from_numpy(x).to(gpu)
    rr = default_radius_range(xdim, ydim)
    total = passes * sch.steps
    quantum = sd.exact_sum_quantum(float(np.abs(x).max()), n) if dtype == np.float64 else 0.0
    outs = []
    for unfused in (False, True):
        st = sd.BatchTrainState(n, c, xdim, ydim, sch, gpu, dtype=xd.dtype)
        st.quantum = quantum
        st.wbuf[0].copy_(torch.from_numpy(w0))
        sd.batch_train_steps(xd, st, 0, total, total, (0.05, 0.01), rr, unfused=unfused)
        w = torch.empty((k, c), dtype=torch.float64, device=gpu)
        sd.batch_train_finish(st, total, total, (0.05, 0.01), rr, w)
        outs.append((w, st.wbuf.clone(), st.ring.clone()))
    (wa, wbuf_a, ring_a), (wb, wbuf_b, ring_b) = outs
    g = total - 1
    assert torch.equal(wbuf_a[g % 2], wbuf_b[g % 2]), "codebook of the last step"
    assert torch.equal(ring_a[g % 3], ring_b[g % 3]), "statistics of the last step"
    assert float(ring_a[(g + 1) % 3].abs().max()) == 0.0, "next statistics buffer not cleared"
    assert torch.equal(wa, wb)
    want = oracle.som_batch_sched(x.astype(np.float64), w0, xdim, ydim, passes, (0.05, 0.01), rr, sch.phases, sch.edges,
                                  quantum=quantum)
    # exact sums + a gain free of library calls: the whole run is the oracle's, bit for bit
    np.testing.assert_array_equal(wa.cpu().numpy(), want)


@pytest.mark.parametrize("c,dtype,pad,grid", [
    (100, np.float32, 0, 10),     # config 4's shape: every kernel of the generic route reads the step's rows where they lie (RowView)
    (100, np.float32, 4, 10),     # the same rows inside a wider matrix (ldx = 104): the gathered copy, as before
    (96, np.float16, 0, 10),      # binary16: packed-K filter + the sums kernel's eight-element vectors on the view
    (72, np.float32, 0, 10),      # three channel chunks
    (48, np.float32, 0, 10),      # <= 64 channels, 100 nodes: the wave-private sums kernels address flat ranges -- gathered
    (40, np.float16, 0, 20),      # config 5's shape: a 400 x 40 table has no wave-private route -- viewed
    (40, np.float32, 0, 20),
])
def test_generic_route_on_row_views_matches_the_oracle(gpu, oracle, c, dtype, pad, grid):
    """Round 6: the launch-per-phase route and the wide one-launch step without the gathered copy of the matrix.  Whole run against
    orc_som_batch_sched on data whose sums are exact (bit for bit), for shapes on both sides of the decision."""
    xdim = ydim = grid
    k, n = grid * grid, 41_003
    sch = MIXED
    x = synth.make_fov_numpy(n, c, seed=61, dtype=np.float32)
    x = (np.round(x.astype(np.float64) * 4096.0) / 4096.0).astype(dtype)
    w0 = _codebook(synth.make_fov_numpy(4 * k, c, seed=62, dtype=np.float64), k, seed=9)
    w0[k - 2] = w0[3]
    host = np.zeros((n, c + pad), dtype=dtype)
    host[:, :c] = x
    xd = torch.from_numpy(host).to(gpu)[:, :c]
    rr = default_radius_range(xdim, ydim)
    total = sch.steps
    st = sd.BatchTrainState(n, c, xdim, ydim, sch, gpu, dtype=xd.dtype)
    st.wbuf[0].copy_(torch.from_numpy(w0))
    sd.batch_train_steps(xd, st, 0, total, total, (0.05, 0.01), rr)
    w = torch.empty((k, c), dtype=torch.float64, device=gpu)
    sd.batch_train_finish(st, total, total, (0.05, 0.01), rr, w)
    want = oracle.som_batch_sched(x.astype(np.float64), w0, xdim, ydim, 1, (0.05, 0.01), rr, sch.phases, sch.edges)
    np.testing.assert_array_equal(w.cpu().numpy(), want)
    # and step by step from another state (the multi-call form: a view needs no state between calls)
    st2 = sd.BatchTrainState(n, c, xdim, ydim, sch, gpu, dtype=xd.dtype)
    st2.wbuf[0].copy_(torch.from_numpy(w0))
    for g in range(total):
        sd.batch_train_steps(xd, st2, g, g + 1, total, (0.05, 0.01), rr)
    w2 = torch.empty((k, c), dtype=torch.float64, device=gpu)
    sd.batch_train_finish(st2, total, total, (0.05, 0.01), rr, w2)
    assert torch.equal(w, w2)


@pytest.mark.parametrize("c,dtype,pad,sch", [
    (2, np.float64, 7, BatchSchedule.equal(16)),     # the case the randomised sweep found (round 6): the wide one-launch step
    (2, np.float32, 7, SMALL_TWO_PHASE),
    (100, np.float32, 0, MIXED),                     # wide step + streamed filter
    (22, np.float32, 0, SMALL_TWO_PHASE),            # the fused 10 x 10 step: windowed and BMU-only steps (no duplicate mask there)
    (22, np.float64, 0, SMALL_TWO_PHASE),
])
def test_all_zero_table_and_codebook_take_the_first_node(gpu, oracle, c, dtype, pad, sch):
    """Every node equals the centring vector and every row sits on it: all terms of the filter's tolerance vanish, while the index
    bits packed into scores of +0 still differ by subnormal steps.  The tolerance has a floor for this (kTolFloor): the rows are
    listed and the exact path takes the FIRST of the equal nodes, as the oracle does."""
    xdim = ydim = 10
    k, n = 100, 3_001
    host = np.zeros((n, c + pad), dtype=dtype)
    xd = torch.from_numpy(host).to(gpu)[:, :c]
    w0 = np.zeros((k, c))
    rr = default_radius_range(xdim, ydim)
    total = sch.steps
    for unfused in (False, True):
        st = sd.BatchTrainState(n, c, xdim, ydim, sch, gpu, dtype=xd.dtype)
        st.wbuf[0].copy_(torch.from_numpy(w0))
        for g in range(total):
            sd.batch_train_steps(xd, st, g, g + 1, total, (0.05, 0.01), rr, unfused=unfused)
            counts = st.ring[g % 3][k * c:].cpu().numpy()
            rows = len(sch.rows_of_step(n, g))
            assert counts[0] == rows and counts[1:].sum() == 0, "step %d (unfused=%s): rows at nodes %s" % (g, unfused, np.nonzero(counts)[0])
        w = torch.empty((k, c), dtype=torch.float64, device=gpu)
        sd.batch_train_finish(st, total, total, (0.05, 0.01), rr, w)
        assert float(w.abs().max()) == 0.0
    # labelling calls on the same table: node 1 (labels are 1-based) for every row
    lab, _ = sd.assign(xd, torch.zeros((k, c), dtype=torch.float64, device=gpu))
    assert int(lab.min()) == 1 and int(lab.max()) == 1
