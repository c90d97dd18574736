"""FOV-sharded batch SOM training: one process per GPU, per-step codebook statistics
all-reduced over RCCL (``torch.distributed`` backend "nccl" on ROCm).

The reference has no analogue (its training is pyFlowSOM's sequential loop,
/root/reference/src/ark/phenotyping/cluster_helpers.py:106-109, "replicas only" in DESIGN.md).
This is the throughput-mode rule BASELINE.json's north_star asks for: pixels never leave their
GPU; the only exchange is one all-reduce of the [K, C+1] binary64 statistics per
mini-batch step (18.4 KB at K=100, C=22 -- latency-bound, xGMI bandwidth is irrelevant).

One pass = ``batch_steps`` mini-batch steps; step g (of G = rlen*batch_steps) uses the local
rows i with i % batch_steps == g % batch_steps:
    labels = BMU(rows, W); S[b] += x_i, n[b] += 1     (pxsom_batch_accumulate: one launch)  -> all-reduce(S, n)
    thr = r0 - (r0-r1) g/G (0.5 once < 1);  alpha = a0 - (a0-a1) g/G
    W_k += (1 - (1-alpha)^den_k) (num_k/den_k - W_k)                      (pxsom_batch_update_prepare:
                                                  also clears S, n and prepares the next step's BMU search)
Oracle of record: oracle/pxsom_oracle.c (orc_som_batch).
"""
from typing import Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def batch_schedule(g: int, total_steps: int, alpha_range: Sequence[float],
                   radius_range: Sequence[float]) -> Tuple[float, float]:
    """(neighbourhood threshold, learning rate) of mini-batch step g -- the online schedule
    sampled at the step's first presentation."""
    a0, a1 = float(alpha_range[0]), float(alpha_range[1])
    r0, r1 = float(radius_range[0]), float(radius_range[1])
    thr = r0 - (r0 - r1) * float(g) / float(total_steps)
    if thr < 1.0:
        thr = 0.5
    alpha = a0 - (a0 - a1) * float(g) / float(total_steps)
    return thr, alpha


class HipKernels:
    """The product kernel set: libpxsom.so through the C ABI (no other implementation ships)."""

    def __init__(self):
        from . import som_device
        self._sd = som_device
        self._ws = None
        self._prepared_for = None

    def accumulate(self, x: torch.Tensor, w: torch.Tensor, labels: torch.Tensor, stats: torch.Tensor,
                   chained: bool = False) -> None:
        """zero(stats); labels = BMU(x, w); stats[label] += [x, 1]  (stats = [K*C sums | K counts] f64).
        ``chained``: the caller guarantees that neither ``w`` nor ``stats`` changed since this object's last
        ``update_prepare`` cleared ``stats`` (then no memset / prep launch precedes the filter)."""
        n, c = x.shape
        if self._ws is None or not self._ws.fits(n, c, w.shape[0]):
            self._ws = self._sd.AssignWorkspace(n, c, w.shape[0], x.device)
            self._prepared_for = None
        # the previous step's update_prepare cleared exactly this buffer for exactly this codebook
        prepared = chained and self._prepared_for == (w.data_ptr(), stats.data_ptr())
        self._prepared_for = None
        self._sd.batch_accumulate(x, w, labels, stats, self._ws, prepared=prepared)

    def batch_update(self, w, xdim, ydim, sums, counts, thr, alpha):
        return self._sd.batch_update(w, xdim, ydim, sums, counts, thr, alpha)

    def update_prepare(self, w, xdim, ydim, stats, thr, alpha, stats_next=None) -> None:
        """Codebook update from ``stats``; the same launch clears ``stats_next`` (the buffer of the next
        accumulate) and, where the shape needs one, a prep launch readies the workspace."""
        self._sd.batch_update_prepare(w, xdim, ydim, stats, thr, alpha, self._ws, stats_next=stats_next)
        cleared = stats_next if stats_next is not None else stats
        self._prepared_for = (w.data_ptr(), cleared.data_ptr())


def _world(group) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


class BatchSOMTrainer:
    """Holds the per-rank buffers of the batch rule so repeated passes allocate nothing."""

    def __init__(self, xdim: int, ydim: int, channels: int, device, batch_steps: int = 64,
                 alpha_range: Sequence[float] = (0.05, 0.01),
                 radius_range: Optional[Sequence[float]] = None, group=None, kernels=None):
        from .flowsom import default_radius_range
        self.xdim, self.ydim, self.k, self.c = int(xdim), int(ydim), int(xdim * ydim), int(channels)
        self.batch_steps = int(batch_steps)
        if self.batch_steps < 1:
            raise ValueError("batch_steps must be >= 1")
        self.alpha_range = tuple(alpha_range)
        self.radius_range = tuple(radius_range) if radius_range is not None else \
            default_radius_range(xdim, ydim)
        self.group = group
        self.kernels = kernels if kernels is not None else HipKernels()
        # statistics: [K*C sums | K counts], all float64 (counts are exact integers): one all-reduce, no
        # conversions.  Two buffers alternate between steps, so the update launch of step g can clear the
        # buffer step g+1 accumulates into while other workgroups still read step g's.
        self._stats_pair = [torch.zeros(self.k * self.c + self.k, dtype=torch.float64, device=device)
                            for _ in range(2)]
        self.stats = self._stats_pair[0]
        self.label_buf = None

    @property
    def sums(self) -> torch.Tensor:
        return self.stats[: self.k * self.c].view(self.k, self.c)

    @property
    def counts(self) -> torch.Tensor:
        return self.stats[self.k * self.c:]

    def step(self, x_local: torch.Tensor, w: torch.Tensor, g: int, total_steps: int,
             chained: bool = False) -> None:
        """One mini-batch step g on this rank's shard; collective when world_size > 1.
        ``chained``: this call directly follows step g-1 of the same ``train`` (nothing touched ``w``
        in between), so the BMU search may reuse what the previous update prepared."""
        m = self.batch_steps
        view = x_local[(g % m)::m]
        nrows = view.shape[0]
        if self.label_buf is None or self.label_buf.numel() < nrows:
            self.label_buf = torch.empty(max(nrows, 1), dtype=torch.int32, device=x_local.device)
        self.stats, stats_next = self._stats_pair[g % 2], self._stats_pair[(g + 1) % 2]
        if chained and hasattr(self.kernels, "update_prepare"):
            self.kernels.accumulate(view, w, self.label_buf, self.stats, chained=True)
        else:
            self.kernels.accumulate(view, w, self.label_buf, self.stats)
        if _world(self.group) > 1:
            dist.all_reduce(self.stats, op=dist.ReduceOp.SUM, group=self.group)
        thr, alpha = batch_schedule(g, total_steps, self.alpha_range, self.radius_range)
        if hasattr(self.kernels, "update_prepare"):
            self.kernels.update_prepare(w, self.xdim, self.ydim, self.stats, thr, alpha, stats_next)
        else:
            self.kernels.batch_update(w, self.xdim, self.ydim, self.sums, self.counts, thr, alpha)

    def train(self, x_local: torch.Tensor, w: torch.Tensor, num_passes: int = 1) -> torch.Tensor:
        """Runs num_passes passes in place on ``w`` [K, C] f64 (identical on every rank)."""
        total = int(num_passes) * self.batch_steps
        for g in range(total):
            self.step(x_local, w, g, total, chained=g > 0)
        return w


def broadcast_codebook(w: torch.Tensor, src: int = 0, group=None) -> torch.Tensor:
    if _world(group) > 1:
        dist.broadcast(w, src=src, group=group)
    return w


def allreduce_cluster_tables(sums: torch.Tensor, counts: torch.Tensor, group=None):
    """K8 across ranks: one sum all-reduce each for the [K, C] f64 sums and [K] i64 counts."""
    if _world(group) > 1:
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
    return sums, counts
