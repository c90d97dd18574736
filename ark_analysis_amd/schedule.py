"""Mini-batch schedules of the batch SOM rule (include/pxsom.h "batch SOM training on a SCHEDULE";
oracle/pxsom_oracle.c orc_som_batch_sched; the reference call the rule stands in for is
PixieSOMCluster.train_som, /root/reference/src/ark/phenotyping/cluster_helpers.py:98-116).

The rows of a pass are dealt into ``phases`` phases (row i: phase ``i % phases``); step g takes the phases
``[edges[g], edges[g+1])`` and its update is taken at the point of the online schedule where the rows presented
before it end.  A step costs the GPU one latency-bound launch whatever its size, so a pass is priced in steps:

* ``BatchSchedule.equal(G)``: G equal steps (rows ``i % G == g``) -- the rule of rounds 1 and 2;
* ``BatchSchedule.two_phase()`` (the default of ``train_mode="batch"``): 6 shrinking steps over the 5/6 of a pass in
  which the neighbourhood radius is >= 1, then 16 steps over the BMU-only tail -- 15 equal ones and a last one five times
  their size.  On the bench workload its mean quantisation error against the online rule's codebook is +0.4 ... +0.6 %
  (two sets of eight seeds, +- 0.1 ... 0.2) where 64 equal steps give +0.6 % and 32 equal steps +1.6 %
  (scripts/study/batch_schedule_scan{2,3,4,5}.py): the tail is a mini-batch k-means whose quality follows the number of
  its iterations, the ordering phase needs few, and the codebook a pass returns carries the sampling noise of its LAST
  mini-batch -- a larger last step buys back what four fewer launches cost (6 + 20 equal tail steps: the same quality).
"""
from typing import Sequence, Tuple, Union

import numpy as np

MAX_STEPS = 256  # include/pxsom.h PXSOM_MAX_SCHED_STEPS


class BatchSchedule:
    def __init__(self, phases: int, edges: Sequence[int]):
        self.phases = int(phases)
        self.edges = tuple(int(e) for e in edges)
        if self.phases < 1 or len(self.edges) < 2 or len(self.edges) - 1 > MAX_STEPS:
            raise ValueError("a schedule has 1 .. %d steps over >= 1 phases" % MAX_STEPS)
        if self.edges[0] != 0 or self.edges[-1] != self.phases or any(b < a for a, b in zip(self.edges, self.edges[1:])):
            raise ValueError("schedule edges must run from 0 to phases without decreasing")

    @property
    def steps(self) -> int:
        return len(self.edges) - 1

    def edges_array(self) -> np.ndarray:
        return np.ascontiguousarray(np.asarray(self.edges, dtype=np.int32))

    def position(self, g: int) -> int:
        """Phases presented before global step g (g counts over all passes)."""
        return (g // self.steps) * self.phases + self.edges[g % self.steps]

    def rows_of_step(self, n: int, g: int) -> np.ndarray:
        """Indices of the rows step g (of a pass) takes out of n rows, ascending."""
        e0, e1 = self.edges[g % self.steps], self.edges[g % self.steps + 1]
        base = np.arange(0, n, self.phases, dtype=np.int64)[:, None] + np.arange(e0, e1, dtype=np.int64)[None, :]
        idx = base.reshape(-1)
        return idx[idx < n]

    def __eq__(self, other):
        return isinstance(other, BatchSchedule) and (self.phases, self.edges) == (other.phases, other.edges)

    def __hash__(self):
        return hash((self.phases, self.edges))

    def __repr__(self):
        return "BatchSchedule(phases=%d, steps=%d)" % (self.phases, self.steps)

    @classmethod
    def equal(cls, steps: int) -> "BatchSchedule":
        steps = int(steps)
        if steps < 1:
            raise ValueError("batch_steps must be a positive integer")
        return cls(steps, range(steps + 1))

    @classmethod
    def two_phase(cls, head_steps: int = 6, tail_steps: int = 16, head_ratio: float = 0.25,
                  tail_phases_per_step: int = 8, last_step_factor: int = 5) -> "BatchSchedule":
        """``head_steps`` steps with geometrically shrinking sizes (last / first = ``head_ratio``) over the first 5/6 of
        the rows -- where the default radius schedule (r0 -> 0) keeps the neighbourhood radius >= 1 for a 10 x 10 map --,
        ``tail_steps`` steps over the last 1/6: equal ones, the last ``last_step_factor`` times their size (the codebook
        a pass returns carries the sampling noise of its last mini-batch)."""
        head_steps, tail_steps, factor = int(head_steps), int(tail_steps), int(last_step_factor)
        if head_steps < 1 or tail_steps < 1 or factor < 1:
            raise ValueError("two_phase needs at least one step in each part")
        unit = int(tail_phases_per_step)
        tail = (tail_steps - 1 + factor) * unit
        phases = 6 * tail
        sizes = np.geomspace(1.0, float(head_ratio), head_steps)
        cum = np.round(np.cumsum(sizes) / sizes.sum() * (phases - tail)).astype(np.int64)
        cum[-1] = phases - tail
        edges = [0] + [int(v) for v in cum] + [phases - tail + (i + 1) * unit for i in range(tail_steps - 1)] + [phases]
        return cls(phases, edges)


DEFAULT = "two-phase"


def resolve(spec: Union[int, str, BatchSchedule, None]) -> BatchSchedule:
    """``batch_steps`` as the plugin API accepts it: an int (equal steps), "two-phase" / None (the default schedule) or
    a BatchSchedule."""
    if isinstance(spec, BatchSchedule):
        return spec
    if spec is None or (isinstance(spec, str) and spec in ("two-phase", "auto")):
        return BatchSchedule.two_phase()
    if isinstance(spec, (int, np.integer)) and not isinstance(spec, bool):
        if int(spec) < 1:
            raise ValueError("batch_steps must be a positive integer")
        return BatchSchedule.equal(int(spec))
    raise ValueError("batch_steps must be a positive integer, 'two-phase' or a BatchSchedule, got %r" % (spec,))
