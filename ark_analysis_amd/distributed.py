"""FOV-sharded batch SOM training: one process per GPU, per-step codebook statistics
all-reduced over RCCL (``torch.distributed`` backend "nccl" on ROCm).

The reference has no analogue (its training is pyFlowSOM's sequential loop,
/root/reference/src/ark/phenotyping/cluster_helpers.py:106-109, "replicas only" in DESIGN.md).
This is the throughput-mode rule BASELINE.json's north_star asks for: pixels never leave their
GPU; the only exchange is one all-reduce of the [K, C+1] binary64 statistics per
mini-batch step (18.4 KB at K=100, C=22 -- latency-bound, xGMI bandwidth is irrelevant).

One pass runs on a schedule (``ark_analysis_amd.schedule.BatchSchedule``): the local rows are dealt into
``phases`` phases (row i: phase i % phases), step g takes the phases [edges[g], edges[g+1]):
    labels = BMU(rows, W); S[b] += x_i, n[b] += 1          -> all-reduce(S, n)
    f = (rows presented before the step) / (rows of the run)
    thr = r0 - (r0-r1) f (0.5 once < 1);  alpha = a0 - (a0-a1) f
    W_k += (1 - (1-alpha)^den_k) (num_k/den_k - W_k)
``batch_steps=G`` (an int) is the equal schedule of rounds 1-2 (rows i % G == g); the default is the two-phase
schedule (6 large steps while the radius is >= 1, 16 in the BMU-only tail, the last one five times the others: the quality of 64 equal
steps in 22 dependent launches).  pxsom_batch_train_sched runs the step loop inside the library: for the
register-resident shapes a step is ONE launch (the update of step g-1 and the codebook preparation sit at the head of
step g's BMU search).  Oracle of record: oracle/pxsom_oracle.c (orc_som_batch_sched).
"""
import time
from typing import Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def batch_schedule(g: int, total_steps: int, alpha_range: Sequence[float],
                   radius_range: Sequence[float]) -> Tuple[float, float]:
    """(neighbourhood threshold, learning rate) at position g / total_steps of the online schedule (equal steps: step
    g of total_steps; a scheduled run: phases presented before the step / phases of the run)."""
    a0, a1 = float(alpha_range[0]), float(alpha_range[1])
    r0, r1 = float(radius_range[0]), float(radius_range[1])
    thr = r0 - (r0 - r1) * float(g) / float(total_steps)
    if thr < 1.0:
        thr = 0.5
    alpha = a0 - (a0 - a1) * float(g) / float(total_steps)
    return thr, alpha


class HipKernels:
    """The product kernel set: libpxsom.so through the C ABI (no other implementation ships).

    Interface the trainer drives (tests/oracle_backend.py holds an oracle-backed stand-in with the same
    methods, for the host logic on CPU):
        begin(x, w, xdim, ydim, schedule)         state for this matrix; W_0 = w
        steps(x, g0, g1, total, alpha, radius)    mini-batch steps [g0, g1) of total = passes x steps, back to back
        ring(g)                                   statistics of step g: what the ranks all-reduce after it
        finish(steps_done, total, alpha, radius, w)   last pending update; w receives the result
    """

    def __init__(self, unfused: bool = False):
        from . import som_device
        self._sd = som_device
        self._state = None
        self.unfused = bool(unfused)

    def absmax(self, x: torch.Tensor) -> float:
        return float(self._sd.absmax(x).item()) if x.shape[0] else 0.0

    def begin(self, x: torch.Tensor, w: torch.Tensor, xdim: int, ydim: int, schedule, group=None, quantum: float = 0.0) -> None:
        n, c = x.shape
        st = self._state
        if st is None or not st.fits(n, c, xdim, ydim, schedule, x.dtype) or st.wbuf.device != x.device:
            st = self._state = self._sd.BatchTrainState(n, c, xdim, ydim, schedule, x.device, dtype=x.dtype)
            self._rings = [st.ring[i] for i in range(3)]     # views made once: ring(g) sits in the per-step loop
        # W_0 stays where the caller holds it: the first run of steps hands it to the library, whose preparing launch copies it into
        # the state (no copy launch in front of the pass)
        self._w0 = w if (w.is_cuda and w.dtype == torch.float64 and w.is_contiguous()) else None
        if self._w0 is None:
            st.wbuf[0].copy_(w)
        st.quantum = float(quantum)
        # The route (one-launch fused step / launch per phase) is a collective decision: a rank with an oddly aligned or
        # empty shard must not part ways with the others.
        # (every rank enters the collective whatever its own flag says: a debug toggle on one rank must not leave the others
        # waiting inside an all-reduce the rest skipped)
        self._unfused_now = self.unfused
        self.route_agreement = None
        if _world(group) > 1:
            # launch-per-phase everywhere only where the ranks DISAGREE (some could take the one-launch 10 x 10 step, some
            # not -- an oddly aligned or empty shard), or where a rank asked for it.  Where no rank can take the fused step
            # (cell SOMs, other grids) nothing is forced: the library picks among the wide one-launch step and the
            # launch-per-phase route per step, which give the same bits (tests/test_gpu_schedule.py,
            # test_wide_bmu_only_steps_match_the_oracle_per_step).
            mine = (not self.unfused) and self._sd.batch_train_fused_route(x, xdim, ydim, st.schedule)
            # one MIN-reduce carries "all", "any" and "nobody asked": [1 if mine] is 1 only when every rank can take the fused step,
            # [-1 if mine] is -1 as soon as ONE rank can, [1 unless this rank asked for the launch-per-phase route]
            flags = torch.tensor([1 if mine else 0, -1 if mine else 0, 0 if self.unfused else 1], dtype=torch.int32,
                                 device=_collective_device(group))
            dist.all_reduce(flags, op=dist.ReduceOp.MIN, group=group)
            all_fused, any_fused, nobody_asked = bool(flags[0].item()), flags[1].item() == -1, bool(flags[2].item())
            self._unfused_now = (not nobody_asked) or (any_fused and not all_fused)
            self.route_agreement = {"all_fused": all_fused, "any_fused": any_fused, "unfused_now": self._unfused_now}

    def steps(self, x, g0: int, g1: int, total: int, alpha_range, radius_range, comm=None) -> None:
        self._sd.batch_train_steps(x, self._state, g0, g1, total, alpha_range, radius_range,
                                   unfused=self._unfused_now, comm=comm, w0=self._w0 if g0 == 0 else None)

    def exchange(self, group=None):
        """The in-library exchange over ``group`` (an RCCL communicator owned by libpxsom, made once per process
        group): with it ``steps(..., comm=...)`` all-reduces every step's statistics itself.  None when the
        group is not RCCL-backed (gloo) or PXSOM_NATIVE_EXCHANGE=0: the trainer then all-reduces ``ring(g)``."""
        return native_exchange(group)

    def ring(self, g: int) -> torch.Tensor:
        return self._rings[g % 3]

    def finish(self, steps_done: int, total: int, alpha_range, radius_range, w: torch.Tensor) -> None:
        self._sd.batch_train_finish(self._state, steps_done, total, alpha_range, radius_range, w)


def _world(group) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


class BatchSOMTrainer:
    """Batch-rule SOM training on this rank's rows; collective when world_size > 1.  One process: the whole
    run is ONE library call (the step loop lives in libpxsom); several ranks: the same call with the library's own
    RCCL exchange behind every step, or one call per step with the packed statistics all-reduced in between.
    ``batch_steps``: an int (equal steps), "two-phase" / None (the default schedule) or a BatchSchedule."""

    def __init__(self, xdim: int, ydim: int, channels: int, device, batch_steps=None,
                 alpha_range: Sequence[float] = (0.05, 0.01),
                 radius_range: Optional[Sequence[float]] = None, group=None, kernels=None):
        from .flowsom import default_radius_range
        from .schedule import resolve
        self.xdim, self.ydim, self.k, self.c = int(xdim), int(ydim), int(xdim * ydim), int(channels)
        self.schedule = resolve(batch_steps)
        self.batch_steps = self.schedule.steps
        # the default was asked for (not a schedule of the caller's): train() may swap it for equal steps on tiny tables
        self._default_schedule = batch_steps is None or (isinstance(batch_steps, str) and batch_steps in ("two-phase", "auto"))
        self.alpha_range = tuple(alpha_range)
        self.radius_range = tuple(radius_range) if radius_range is not None else \
            default_radius_range(xdim, ydim)
        self.group = group
        self.kernels = kernels if kernels is not None else HipKernels()

    def _sum_quantum(self, x_local: torch.Tensor) -> float:
        """binary64 rows (what the drop-in classes hand over) train reproducibly: every value joins the per-BMU sums
        rounded to a power of two q chosen so that all partial sums are exact -- the statistics, hence the whole run, are
        then independent of summation order and workgroup count for a given sharding (a different world size deals the rows
        into different steps: another run of the rule, not another rounding) (include/pxsom.h "Reproducible statistics";
        reference property: same-seed retraining gives the same weights, tests/phenotyping/cluster_helpers_test.py:323-332).
        q follows from the job's largest |value| and the most rows a step holds over all ranks; 0 for other dtypes."""
        if x_local.dtype != torch.float64:
            return 0.0
        from . import som_device
        world = _world(self.group)
        vmax, n_total = self.kernels.absmax(x_local), int(x_local.shape[0])
        if world > 1:
            t = torch.tensor([vmax, float(n_total)], dtype=torch.float64, device=_collective_device(self.group))
            dist.all_reduce(t[:1], op=dist.ReduceOp.MAX, group=self.group)
            dist.all_reduce(t[1:], op=dist.ReduceOp.SUM, group=self.group)
            vmax, n_total = float(t[0].item()), int(t[1].item())
        widest = max(b - a for a, b in zip(self.schedule.edges, self.schedule.edges[1:]))
        return som_device.exact_sum_quantum(vmax, (n_total // self.schedule.phases + world) * max(widest, 1))

    def train(self, x_local: torch.Tensor, w: torch.Tensor, num_passes: int = 1, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Runs num_passes passes from ``w`` [K, C] f64 (identical on every rank); the result replaces ``w``, or goes to ``out``
        (same shape and dtype; ``w`` then keeps the first codebook -- a caller that trains from the same start again needs no copy)."""
        w_first = w
        if out is not None:
            if tuple(out.shape) != tuple(w.shape) or out.dtype != w.dtype or out.device != w.device:
                raise ValueError("out does not match the codebook")
        if self._default_schedule:
            # The two-phase default deals the rows into 960 phases: with fewer than a few rows per phase its tail steps hold
            # next to nothing (below 960 rows: nothing at all, and the rows past index 800 would never be presented).  Small
            # tables -- a cell SOM over a few thousand cells -- take equal steps instead, as rounds 1-2 did: at most 64,
            # at least one row per node and step where the table allows it.  Decided on the JOB's row count (all ranks).
            from .schedule import BatchSchedule
            n_total = int(x_local.shape[0])
            if _world(self.group) > 1:
                t = torch.tensor([float(n_total)], dtype=torch.float64, device=_collective_device(self.group))
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
                n_total = int(t.item())
            two_phase = BatchSchedule.two_phase()
            want = two_phase if n_total >= 8 * two_phase.phases else BatchSchedule.equal(max(1, min(64, n_total // max(self.k, 1))))
            if want != self.schedule:
                self.schedule, self.batch_steps = want, want.steps
        total = int(num_passes) * self.batch_steps
        if total < 1:
            if out is not None:
                out.copy_(w)
                return out
            return w
        if x_local.shape[1] != self.c or tuple(w.shape) != (self.k, self.c):
            raise ValueError(f"matrix [{tuple(x_local.shape)}] / codebook [{tuple(w.shape)}] do not match "
                             f"the trainer's {self.k} nodes x {self.c} channels")
        kern = self.kernels
        quantum = self._sum_quantum(x_local)
        if isinstance(kern, HipKernels):
            kern.begin(x_local, w, self.xdim, self.ydim, self.schedule, group=self.group, quantum=quantum)
        else:
            kern.begin(x_local, w, self.xdim, self.ydim, self.schedule, quantum=quantum)
        if _world(self.group) > 1:
            comm = kern.exchange(self.group) if hasattr(kern, "exchange") else None
            if comm is not None:     # step launches and their all-reduces back to back on one stream, one call
                kern.steps(x_local, 0, total, total, self.alpha_range, self.radius_range, comm=comm)
                if not _exchange_completed(comm, self.group):
                    # A peer did not arrive at an exchange in time (the one-shot peer-to-peer route bounds its wait: the late
                    # rank's statistics turned to NaN there and that step was dropped on that rank only -- the replicas have
                    # parted).  Agreed by all ranks: the route is retired and the pass runs again from W_0 (``w`` still holds
                    # it: finish has not run) with the statistics all-reduced through torch.distributed.
                    _retire_native_exchange(self.group, "a peer was late for an exchange")
                    kern.begin(x_local, w, self.xdim, self.ydim, self.schedule, group=self.group, quantum=quantum) \
                        if isinstance(kern, HipKernels) else kern.begin(x_local, w, self.xdim, self.ydim, self.schedule, quantum=quantum)
                    comm = None
            if comm is None:
                for g in range(total):
                    kern.steps(x_local, g, g + 1, total, self.alpha_range, self.radius_range)
                    all_reduce_(kern.ring(g), group=self.group)
        else:
            kern.steps(x_local, 0, total, total, self.alpha_range, self.radius_range)
        w = w_first if out is None else out
        kern.finish(total, total, self.alpha_range, self.radius_range, w)
        if _world(self.group) > 1:
            # every rank applied the same all-reduced statistics, so the replicas are equal already; the broadcast makes
            # that unconditional (rank 0's file is the codebook of record) for the price of one 18 KB message
            broadcast_codebook(w, dist.get_global_rank(self.group, 0) if self.group is not None else 0, self.group)
        return w


_native_comms = {}   # global ranks of a process group -> RankComm | None (decided once per group, the same on every rank)


def _close_native_comms():
    for comm in list(_native_comms.values()):
        if comm is not None:
            try:
                comm.close()
            except Exception:   # noqa: BLE001 -- interpreter shutdown: the runtime may be gone already
                pass
    _native_comms.clear()


import atexit  # noqa: E402
atexit.register(_close_native_comms)


def _all_ranks_ok(ok: bool, group) -> bool:
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=_collective_device(group))
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return bool(flag.item())


def _exchange_verified(comm, group) -> bool:
    """One all-reduce of a known payload through a communicator that was just made, before any training step depends on it:
    every rank contributes ``j * (rank + 1) + rank / 4`` (exact in binary64, so the sum is the same bits in any order) and
    checks the closed form.  The verdict is agreed (MIN over the ranks): a route whose first run on this hardware returns
    anything else -- or raises -- is dropped by all ranks together, and the caller falls back."""
    ok = True
    try:
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        dev = torch.device("cuda", torch.cuda.current_device())
        base = torch.arange(2300, dtype=torch.float64, device=dev)
        got = base * float(rank + 1) + 0.25 * rank
        comm.allreduce_sum(got)
        torch.cuda.synchronize()
        want = base * (world * (world + 1) / 2.0) + 0.25 * (world * (world - 1) / 2.0)
        ok = bool(torch.equal(got, want))
    except Exception:      # noqa: BLE001 -- agreed below
        ok = False
    return _all_ranks_ok(ok, group)


def _exchange_completed(comm, group) -> bool:
    """After a run of steps through an in-library communicator: did every exchange complete on every rank?  Only the
    peer-to-peer route can say no (``error_epoch``: the first exchange a peer missed); the verdict is agreed (MIN)."""
    ok = True
    if hasattr(comm, "error_epoch"):
        try:
            torch.cuda.synchronize()
            ok = comm.error_epoch() == 0
        except Exception:      # noqa: BLE001 -- agreed below
            ok = False
    return _all_ranks_ok(ok, group)


def _retire_native_exchange(group, why: str) -> None:
    """Drops the group's in-library communicator for the rest of the process (every rank calls this at the same point)."""
    import warnings
    key = _group_key(group)
    comm = _native_comms.get(key)
    if comm is not None:
        try:
            comm.close()
        except Exception:      # noqa: BLE001
            pass
    _native_comms[key] = None
    if dist.get_rank(group) == 0:
        warnings.warn("in-library exchange retired (%s): all-reducing through torch.distributed" % why)


def _group_key(group):
    """What identifies a process group for the communicator cache: its global ranks (an ``id()`` can be reused once the
    group object is collected)."""
    return tuple(sorted(dist.get_process_group_ranks(group))) if group is not None else None


# Which in-library exchange a multi-rank job takes:
#   "auto" (default)  both routes are made and checked; each runs a short timed series of the rule's own all-reduce ([K C + K]
#                     binary64 words); the peer-to-peer route INSIDE the step launches is taken when its one-launch form measured
#                     at most FLIP_RATIO of RCCL's time per exchange (MAX over the ranks: every rank computes the same verdict),
#                     RCCL otherwise.  A route that cannot be made, fails its first checked exchange or times out is not a candidate.
#   "rccl" | "p2p" | "fused"   that route (falling back as before when it cannot be made).
# Set by set_exchange_route() or PXSOM_EXCHANGE in the environment of every rank.
FLIP_RATIO = 0.7
_route = None
exchange_report = {}      # group key -> what was measured and chosen (bench.py prints it)


def set_exchange_route(route: Optional[str]) -> None:
    """"auto" | "rccl" | "p2p" | "fused" | None (= environment / default).  Call it on every rank, before the first training run."""
    global _route
    if route not in (None, "auto", "rccl", "p2p", "fused"):
        raise ValueError("exchange route: auto, rccl, p2p or fused")
    _route = route


def _exchange_route() -> str:
    import os
    route = _route or os.environ.get("PXSOM_EXCHANGE", "auto")
    return route if route in ("auto", "rccl", "p2p", "fused") else "auto"


def _make_p2p(group):
    """A connected, checked peer-to-peer communicator, or (None, why) -- agreed by all ranks."""
    from . import som_device
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    comm, err = None, None
    try:
        comm = som_device.P2PComm(world, rank, 1 << 18)
    except Exception as e:          # noqa: BLE001 -- agreed below: all ranks fall back together
        err = e
    if _all_ranks_ok(comm is not None, group):
        box = [None] * world
        dist.all_gather_object(box, comm.local_handle, group=group)
        try:
            comm.connect(box)
        except Exception as e:      # noqa: BLE001
            err = e
            comm.close()
            comm = None
    elif comm is not None:
        comm.close()
        comm = None
    usable = _all_ranks_ok(comm is not None, group)
    if usable and not _exchange_verified(comm, group):
        usable, err = False, "its first all-reduce did not return the expected sums"
    if usable and not _all_ranks_ok(comm.error_epoch() == 0, group):
        usable, err = False, "a peer was late for its first all-reduce"
    if not usable:
        if comm is not None:
            comm.close()
        return None, err
    return comm, None


def _make_rccl(group):
    from . import som_device
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    comm, uid, err = None, None, None
    try:
        som_device.RankComm.bind()
        if rank == 0:
            uid = som_device.RankComm.unique_id()
    except Exception as e:          # agreed below: all ranks fall back together
        err = e
    if _all_ranks_ok(err is None, group):
        box = [uid]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        try:
            comm = som_device.RankComm(box[0], world, rank)
        except Exception as e:      # noqa: BLE001
            err = e
        usable = _all_ranks_ok(comm is not None, group)
        if usable and not _exchange_verified(comm, group):
            usable, err = False, "its first all-reduce did not return the expected sums"
        if not usable:
            if comm is not None:
                comm.close()
            comm = None
    return comm, err


def _exchange_us(comm, group, words: int = 2300, reps: int = 40) -> float:
    """Microseconds per all-reduce of ``words`` binary64 values through ``comm``, back to back on this stream: MAX over the ranks
    (the same number on every rank); inf where an exchange failed."""
    dev = torch.device("cuda", torch.cuda.current_device())
    probe = torch.zeros(words, dtype=torch.float64, device=dev)
    ok = True
    us = float("inf")
    try:
        for _ in range(6):
            comm.allreduce_sum(probe)
        torch.cuda.synchronize()
        dist.barrier(group=group)
        t0 = time.perf_counter()
        for _ in range(reps):
            comm.allreduce_sum(probe)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / reps * 1e6
        if hasattr(comm, "error_epoch") and comm.error_epoch() != 0:
            ok = False
        if not bool(torch.isfinite(probe).all().item()):
            ok = False
    except Exception:      # noqa: BLE001 -- agreed below
        ok = False
    t = torch.tensor([us if ok else float("inf")], dtype=torch.float64, device=_collective_device(group))
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def native_exchange(group=None):
    """libpxsom's own communicator over the ranks of ``group`` (see HipKernels.exchange): RCCL, or the one-shot peer-to-peer
    exchange over HIP IPC blocks -- on the fused 10 x 10 step inside the step launches.  Collective: every rank of the group calls
    it at the same point.  Every phase is agreed, so that a rank that cannot take part (no RCCL to bind, a block that cannot be
    mapped) never leaves the others waiting.  The route: see ``set_exchange_route``; what was measured and chosen is kept in
    ``exchange_report``."""
    import os
    import warnings
    key = _group_key(group)
    if key in _native_comms:
        return _native_comms[key]
    wanted = os.environ.get("PXSOM_NATIVE_EXCHANGE", "1")     # "0": never; "force": on any backend (tests: gloo group
    route = _exchange_route()                                  # + a stand-in collective library)
    rank = dist.get_rank(group)
    report = {"route_asked": route}
    comm = None
    if wanted != "0" and torch.cuda.is_available():
        on_rccl_backend = dist.get_backend(group) == "nccl"
        p2p = rccl = None
        if route in ("p2p", "fused") or (route == "auto" and on_rccl_backend):
            p2p, why = _make_p2p(group)
            if p2p is None:
                report["p2p"] = "unavailable: %s" % why
                if rank == 0 and route != "auto":
                    warnings.warn("peer-to-peer exchange unavailable (%s): falling back" % why)
        if route in ("p2p", "fused") and p2p is not None:
            comm = p2p
            comm.set_fused(route == "fused")
        elif on_rccl_backend or wanted == "force":
            rccl, why = _make_rccl(group)
            if rccl is None:
                report["rccl"] = "unavailable: %s" % why
            if rccl is not None and p2p is not None:          # auto: both work -- the faster one, by the rule of FLIP_RATIO
                report["rccl_us_per_exchange"] = round(_exchange_us(rccl, group), 2)
                report["p2p_us_per_exchange"] = round(_exchange_us(p2p, group), 2)
                take_p2p = report["p2p_us_per_exchange"] <= FLIP_RATIO * report["rccl_us_per_exchange"]
                report["rule"] = "peer-to-peer inside the step launches when its one-launch exchange takes <= %.2f x RCCL's" % FLIP_RATIO
                comm, other = (p2p, rccl) if take_p2p else (rccl, p2p)
                other.close()
                if take_p2p:
                    comm.set_fused(True)
            elif rccl is not None:
                comm = rccl
            elif p2p is not None:                             # auto without a working RCCL binding
                comm = p2p
                comm.set_fused(True)
            if comm is None and rank == 0:
                warnings.warn("in-library exchange unavailable (%s): all-reducing through torch.distributed" % why)
    report["route_taken"] = ("torch.distributed" if comm is None else
                             ("p2p-fused" if getattr(comm, "fused", False) else "p2p") if hasattr(comm, "error_epoch") else "rccl")
    exchange_report[key] = report
    _native_comms[key] = comm
    return comm


# ---- rank context of the drop-in pipeline --------------------------------------------------------------------
# Under ``torchrun`` (one process per GPU) the pipeline functions shard their FOV lists by rank; without a
# process group everything below degenerates to rank 0 of 1 and costs nothing.

def init_from_env() -> Tuple[int, int]:
    """(rank, world).  Joins the job ``torch.distributed.run`` started (RANK / WORLD_SIZE / MASTER_* in the
    environment) if no process group exists yet: backend "nccl" (= RCCL over xGMI) with a HIP device, else
    "gloo".  A plain ``python`` process stays (0, 1)."""
    import os
    if not dist.is_available():
        return 0, 1
    if not dist.is_initialized():
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if world <= 1 or "RANK" not in os.environ:
            return 0, 1
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        rank = int(os.environ["RANK"])
        if torch.cuda.is_available():
            local = int(os.environ.get("LOCAL_RANK", str(rank % max(1, torch.cuda.device_count()))))
            torch.cuda.set_device(local)
            dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    return dist.get_rank(), dist.get_world_size()


def context() -> Tuple[int, int]:
    """(rank, world) of the current process group; (0, 1) without one."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard(items, rank: Optional[int] = None, world: Optional[int] = None) -> list:
    """This rank's share of ``items`` (round robin over a list every rank holds in the same order)."""
    if rank is None or world is None:
        rank, world = context()
    return list(items)[rank::world]


def barrier() -> None:
    if context()[1] > 1:
        dist.barrier()


def _collective_device(group=None) -> torch.device:
    """Where tensors handed to a collective of ``group`` must live: HBM for RCCL, host memory for gloo."""
    if dist.get_backend(group) == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def broadcast_object(obj, src: int = 0):
    """``obj`` of rank ``src`` on every rank (small host objects: codebooks, FOV lists)."""
    if context()[1] <= 1:
        return obj
    box = [obj]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def on_rank0(fn):
    """``fn()`` of rank 0 on every rank.  An exception on rank 0 travels instead of the result and is re-raised by
    EVERY rank: work that only rank 0 does in front of a collective must not leave the others waiting in it."""
    rank, world = context()
    if world <= 1:
        return fn()
    box = None
    if rank == 0:
        try:
            box = (True, fn())
        except Exception as e:      # noqa: BLE001 -- re-raised below, on every rank
            box = (False, e)
    ok, payload = broadcast_object(box, 0)
    if not ok:
        raise payload
    return payload


def allgather_objects(obj) -> list:
    """[obj of rank 0, obj of rank 1, ...] on every rank."""
    world = context()[1]
    if world <= 1:
        return [obj]
    out = [None] * world
    dist.all_gather_object(out, obj)
    return out


def allreduce_sum_numpy(arr):
    """Element-wise sum over ranks of a float64 / int64 numpy array (returned as a new array)."""
    import numpy as np
    if context()[1] <= 1:
        return np.array(arr, copy=True)
    t = torch.from_numpy(np.ascontiguousarray(arr)).to(_collective_device())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def _staged(t: torch.Tensor, group, fn) -> None:
    """Runs the in-place collective ``fn`` on ``t`` where the group's backend can see it: RCCL takes the HBM tensor as it
    is, a gloo group (the CPU test suite; a dry run of the multi-rank code on one GPU) takes a host copy."""
    dev = _collective_device(group)
    if t.device.type == dev.type:
        fn(t)
    else:
        h = t.to(dev)
        fn(h)
        t.copy_(h)


def all_reduce_(t: torch.Tensor, op=None, group=None) -> torch.Tensor:
    if _world(group) > 1:
        _staged(t, group, lambda v: dist.all_reduce(v, op=op if op is not None else dist.ReduceOp.SUM, group=group))
    return t


def broadcast_codebook(w: torch.Tensor, src: int = 0, group=None) -> torch.Tensor:
    if _world(group) > 1:
        _staged(w, group, lambda v: dist.broadcast(v, src=src, group=group))
    return w


def allreduce_cluster_tables(sums: torch.Tensor, counts: torch.Tensor, group=None):
    """K8 across ranks: one sum all-reduce each for the [K, C] f64 sums and [K] i64 counts."""
    all_reduce_(sums, group=group)
    all_reduce_(counts, group=group)
    return sums, counts
