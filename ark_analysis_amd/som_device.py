"""Device-level SOM operations: thin, typed wrappers over the C ABI working on torch tensors.

Tensors are only device-memory holders here (``data_ptr()`` goes across the ABI); every
function launches on torch's current HIP stream and returns without synchronising.
"""
from typing import Optional, Tuple

import numpy as np
import torch

from . import _capi


def _matrix_args(x: torch.Tensor) -> Tuple[int, int, int, int]:
    if x.dim() != 2:
        raise ValueError("pixel matrix must be 2-D [rows, channels]")
    if not x.is_cuda:
        raise ValueError("pixel matrix must live in HBM (a cuda/HIP tensor)")
    n, c = x.shape
    if n == 0:                         # an empty shard (more ranks than rows): no element, no stride to check
        return 0, c, c, _capi.dtype_code(x)
    if c > 1 and x.stride(1) != 1:     # (a single column has no second stride to speak of: torch reports anything)
        raise ValueError("pixel matrix rows must be contiguous (stride(1) == 1)")
    ldx = x.stride(0) if n > 1 else max(c, x.stride(0))
    return n, c, ldx, _capi.dtype_code(x)


def _codebook(w: torch.Tensor) -> torch.Tensor:
    if w.dtype != torch.float64 or not w.is_cuda or not w.is_contiguous() or w.dim() != 2:
        raise ValueError("codebook must be a contiguous float64 [K, C] HBM tensor")
    return w


class AssignWorkspace:
    """Scratch for pxsom_assign, reusable across calls of the same (n_max, c, k)."""

    def __init__(self, n_max: int, c: int, k: int, device):
        self.bytes = _capi.lib().pxsom_assign_workspace_bytes(int(n_max), int(c), int(k))
        if self.bytes == 0:
            raise _capi.PxsomError(f"unsupported assign shape n={n_max} c={c} k={k}")
        self.n_max, self.c, self.k = int(n_max), int(c), int(k)
        self.buf = torch.empty(self.bytes, dtype=torch.uint8, device=device)

    def fits(self, n: int, c: int, k: int) -> bool:
        return c == self.c and k == self.k and n <= self.n_max


def assign(x: torch.Tensor, w: torch.Tensor, labels: Optional[torch.Tensor] = None,
           dists: Optional[torch.Tensor] = None, want_dists: bool = False,
           workspace: Optional[AssignWorkspace] = None, screen_all_lists: bool = False) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """BMU labels (int32, 1-based) of every row of ``x`` against codebook ``w`` [K, C] f64.  ``screen_all_lists``: every list of
    rows for the exact path takes the long-list (screened) kernel whatever its length (PXSOM_ASSIGN_SCREEN_ALL_LISTS; same labels)."""
    n, c, ldx, dt = _matrix_args(x)
    w = _codebook(w)
    k = w.shape[0]
    if w.shape[1] != c:
        raise ValueError(f"codebook has {w.shape[1]} channels, matrix has {c}")
    if labels is None:
        labels = torch.empty(n, dtype=torch.int32, device=x.device)
    if want_dists and dists is None:
        dists = torch.empty(n, dtype=torch.float64, device=x.device)
    if workspace is None or not workspace.fits(n, c, k):
        workspace = AssignWorkspace(n, c, k, x.device)
    rc = _capi.lib().pxsom_assign_ex(x.data_ptr(), n, c, ldx, dt, w.data_ptr(), k, labels.data_ptr(),
                                     dists.data_ptr() if dists is not None else None,
                                     workspace.buf.data_ptr(), workspace.bytes, ASSIGN_SCREEN_ALL_LISTS if screen_all_lists else 0,
                                     _capi.stream_ptr())
    _capi.check(rc, "pxsom_assign_ex")
    assign.last_workspace = workspace
    return labels, dists


def last_exact_rows(workspace: AssignWorkspace) -> int:
    import ctypes
    out = ctypes.c_int64(0)
    _capi.check(_capi.lib().pxsom_assign_last_exact_rows(workspace.buf.data_ptr() + getattr(workspace, "assign_offset", 0), _capi.stream_ptr(),
                                                         ctypes.byref(out)),
                "pxsom_assign_last_exact_rows")
    return int(out.value)


def cluster_sums(x: torch.Tensor, labels: torch.Tensor, k: int,
                 sums: Optional[torch.Tensor] = None, counts: Optional[torch.Tensor] = None):
    """Adds per-label channel sums [k, C] f64 and counts [k] i64 of the rows of ``x``."""
    n, c, ldx, dt = _matrix_args(x)
    if labels.dtype != torch.int32 or labels.numel() != n or not labels.is_contiguous():
        raise ValueError("labels must be a contiguous int32 vector with one entry per row")
    if sums is None:
        sums = torch.zeros((k, c), dtype=torch.float64, device=x.device)
    if counts is None:
        counts = torch.zeros(k, dtype=torch.int64, device=x.device)
    rc = _capi.lib().pxsom_cluster_sums(x.data_ptr(), n, c, ldx, dt, labels.data_ptr(), int(k),
                                        sums.data_ptr(), counts.data_ptr(), _capi.stream_ptr())
    _capi.check(rc, "pxsom_cluster_sums")
    return sums, counts


TRAIN_UNFUSED = 1  # include/pxsom.h PXSOM_TRAIN_UNFUSED
ASSIGN_SCREEN_ALL_LISTS = 1  # include/pxsom.h PXSOM_ASSIGN_SCREEN_ALL_LISTS


class BatchTrainState:
    """Caller-owned state of ``pxsom_batch_train_sched``: the codebook twin buffer ``wbuf`` [2, K, C], the
    rotating statistics ``ring`` [3, K*(C+1)] (float64; ``ring[g % 3]`` is what a multi-rank job all-reduces
    after step g) and the scratch workspace for ``n`` training rows of ``dtype`` (default: the widest, so that the
    state fits any matrix) on ``schedule`` (an int: that many equal steps per pass)."""

    def __init__(self, n: int, c: int, xdim: int, ydim: int, schedule, device, dtype=torch.float64):
        from .schedule import resolve
        self.n, self.c, self.xdim, self.ydim = int(n), int(c), int(xdim), int(ydim)
        self.k, self.schedule, self.dtype = self.xdim * self.ydim, resolve(schedule), dtype
        self.batch_steps = self.schedule.steps
        self.edges = self.schedule.edges_array()            # host array handed to every call (kept alive here)
        self.ws_bytes = _capi.lib().pxsom_batch_train_sched_workspace_bytes(
            self.n, self.c, self.k, _capi.dtype_code(torch.empty(0, dtype=dtype)), self.schedule.phases,
            self.edges.ctypes.data, self.schedule.steps)
        if self.ws_bytes == 0:
            raise _capi.PxsomError(f"unsupported batch-training shape n={n} c={c} k={self.k}")
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=device)
        self.wbuf = torch.empty((2, self.k, self.c), dtype=torch.float64, device=device)
        self.ring = torch.zeros((3, self.k * (self.c + 1)), dtype=torch.float64, device=device)
        self.quantum = 0.0      # > 0: binary64 rows join the statistics rounded to its multiples (exact, order-free sums)

    def fits(self, n: int, c: int, xdim: int, ydim: int, schedule, dtype=None) -> bool:
        from .schedule import resolve
        return (c == self.c and xdim == self.xdim and ydim == self.ydim and resolve(schedule) == self.schedule
                and n <= self.n and (dtype is None or torch.empty(0, dtype=dtype).element_size()
                                     <= torch.empty(0, dtype=self.dtype).element_size()))


def absmax(x: torch.Tensor) -> torch.Tensor:
    """max |x| over the finite entries of a matrix, as a 1-element float64 HBM tensor (0 when there is none)."""
    n, c, ldx, dt = _matrix_args(x)
    out = torch.empty(1, dtype=torch.float64, device=x.device)
    _capi.check(_capi.lib().pxsom_absmax(x.data_ptr(), n, c, ldx, dt, out.data_ptr(), _capi.stream_ptr()), "pxsom_absmax")
    return out


def exact_sum_quantum(value_bound: float, rows_bound: int) -> float:
    """The power of two q for which sums of at most ``rows_bound`` multiples of q below ``value_bound`` are exact in binary64
    (include/pxsom.h "Reproducible statistics"); 0 for all-zero data.  Host arithmetic, no GPU."""
    return float(_capi.lib().pxsom_exact_sum_quantum(float(value_bound), int(rows_bound)))


def batch_train_fused_route(x: torch.Tensor, xdim: int, ydim: int, schedule) -> bool:
    """Whether the steps of this matrix take the one-launch fused kernel (a multi-rank job agrees on the route before
    it starts: include/pxsom.h pxsom_batch_train_fused_route)."""
    from .schedule import resolve
    n, c, ldx, dt = _matrix_args(x)
    return bool(_capi.lib().pxsom_batch_train_fused_route(x.data_ptr(), c, ldx, dt, int(xdim), int(ydim),
                                                          resolve(schedule).phases))


def batch_train_steps(x: torch.Tensor, state: BatchTrainState, g_begin: int, g_end: int, total_steps: int,
                      alpha_range, radius_range, unfused: bool = False, comm: "RankComm" = None,
                      w0: Optional[torch.Tensor] = None) -> None:
    """Mini-batch steps [g_begin, g_end) of a batch training run of ``total_steps`` = passes x steps per pass, launched
    back to back by the library (``state.wbuf[0]`` holds W_0 before step 0; see include/pxsom.h).  ``comm``: the
    statistics of every step are sum-all-reduced over its ranks right behind the step's launch (every rank makes the
    same call).  ``w0`` (with ``g_begin == 0``): the run's first codebook where the caller holds it -- the launch that prepares the
    run copies it into ``state.wbuf[0]`` (pxsom_batch_train_sched_from: no copy launch in front of the pass)."""
    n, c, ldx, dt = _matrix_args(x)
    if w0 is not None:
        w0 = _codebook(w0)
        if tuple(w0.shape) != (state.xdim * state.ydim, c):
            raise ValueError("w0 does not match the state's codebook")
    if not state.fits(n, c, state.xdim, state.ydim, state.schedule, x.dtype):
        raise ValueError("batch-training state does not fit this matrix")
    sch = state.schedule
    if int(total_steps) % sch.steps:
        raise ValueError("total_steps must be a whole number of passes")
    rc = _capi.lib().pxsom_batch_train_sched_from(
        x.data_ptr(), n, c, ldx, dt, w0.data_ptr() if (w0 is not None and int(g_begin) == 0) else None,
        state.wbuf.data_ptr(), state.ring.data_ptr(), state.xdim, state.ydim,
        sch.phases, state.edges.ctypes.data, sch.steps, int(g_begin), int(g_end), int(total_steps) // sch.steps,
        float(alpha_range[0]), float(alpha_range[1]), float(radius_range[0]), float(radius_range[1]), float(state.quantum),
        state.ws.data_ptr(), state.ws_bytes, TRAIN_UNFUSED if unfused else 0,
        comm.handle if comm is not None else None, _capi.stream_ptr())
    _capi.check(rc, "pxsom_batch_train_sched_from")


COMM_ID_BYTES = 128  # include/pxsom.h PXSOM_COMM_ID_BYTES


def _torch_rccl_path() -> str:
    """The librccl.so this process already uses: PyTorch's own copy (binding a second RCCL next to it is
    what pxsom_comm_bind exists to avoid); empty = let the library fall back to the system one."""
    import os
    override = os.environ.get("PXSOM_RCCL_LIBRARY")     # another build of the collective library (tests: a stand-in)
    if override:
        return override
    path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    return path if os.path.exists(path) else ""


class RankComm:
    """RCCL communicator owned by libpxsom (pxsom_comm_*): what the in-library exchange of a multi-rank batch
    run reduces over.  ``RankComm.unique_id()`` on one rank, the 128 bytes handed to the others by the
    launcher's channel, then ``RankComm(id, nranks, rank)`` on every rank (collective) with its device current."""

    def __init__(self, uid: bytes, nranks: int, rank: int):
        import ctypes
        self.bind()
        if len(uid) != COMM_ID_BYTES:
            raise ValueError("communicator id must be %d bytes" % COMM_ID_BYTES)
        box = ctypes.c_void_p()
        buf = ctypes.create_string_buffer(bytes(uid), COMM_ID_BYTES)
        rc = _capi.lib().pxsom_comm_create(ctypes.cast(buf, ctypes.c_void_p), COMM_ID_BYTES, int(nranks), int(rank),
                                           ctypes.byref(box))
        _capi.check(rc, "pxsom_comm_create")
        self.handle = box
        self.nranks, self.rank = int(nranks), int(rank)

    @staticmethod
    def bind() -> None:
        rc = _capi.lib().pxsom_comm_bind(_torch_rccl_path().encode())
        _capi.check(rc, "pxsom_comm_bind")

    @staticmethod
    def unique_id() -> bytes:
        import ctypes
        RankComm.bind()
        buf = ctypes.create_string_buffer(COMM_ID_BYTES)
        rc = _capi.lib().pxsom_comm_unique_id(ctypes.cast(buf, ctypes.c_void_p), COMM_ID_BYTES)
        _capi.check(rc, "pxsom_comm_unique_id")
        return buf.raw

    def allreduce_sum(self, t: torch.Tensor) -> None:
        if t.dtype != torch.float64 or not t.is_contiguous():
            raise ValueError("the exchange reduces contiguous float64 buffers")
        rc = _capi.lib().pxsom_comm_allreduce_sum_f64(self.handle, t.data_ptr(), t.numel(), _capi.stream_ptr())
        _capi.check(rc, "pxsom_comm_allreduce_sum_f64")

    def close(self) -> None:
        if self.handle is not None:
            h, self.handle = self.handle, None
            _capi.check(_capi.lib().pxsom_comm_destroy(h), "pxsom_comm_destroy")


P2P_HANDLE_BYTES = 64  # include/pxsom.h PXSOM_P2P_HANDLE_BYTES


class P2PComm:
    """One-shot peer-to-peer exchange owned by libpxsom (pxsom_comm_p2p_*): every rank's block of device memory is mapped
    by the others through HIP IPC, an all-reduce is one launch per rank and gives bit-identical sums on all ranks.  Ranks
    may share a device.  Two phases, so that a launcher can agree on each before the next: the constructor allocates the
    own block (``local_handle``: 64 bytes), ``connect`` takes the handles of all ranks in rank order (``gather``: a
    callable doing both in one go, e.g. over ``torch.distributed.all_gather_object``).  Same interface as RankComm."""

    def __init__(self, nranks: int, rank: int, max_count: int, gather=None):
        import ctypes
        box = ctypes.c_void_p()
        _capi.check(_capi.lib().pxsom_comm_p2p_create(int(nranks), int(rank), int(max_count), ctypes.byref(box)),
                    "pxsom_comm_p2p_create")
        self.handle = box
        self.nranks, self.rank, self.max_count = int(nranks), int(rank), int(max_count)
        self.fused = False
        mine = ctypes.create_string_buffer(P2P_HANDLE_BYTES)
        _capi.check(_capi.lib().pxsom_comm_p2p_handle(self.handle, ctypes.cast(mine, ctypes.c_void_p), P2P_HANDLE_BYTES),
                    "pxsom_comm_p2p_handle")
        self.local_handle = mine.raw          # 64 bytes: what the other ranks need to map this rank's block
        if gather is not None:
            self.connect(gather(self.local_handle))

    def connect(self, handles) -> None:
        """Maps the blocks of all ranks (``handles``: every rank's ``local_handle``, rank order)."""
        import ctypes
        handles = list(handles)
        if len(handles) != self.nranks or any(len(h) != P2P_HANDLE_BYTES for h in handles):
            raise ValueError("one %d-byte handle per rank, in rank order" % P2P_HANDLE_BYTES)
        blob = ctypes.create_string_buffer(b"".join(handles), P2P_HANDLE_BYTES * self.nranks)
        _capi.check(_capi.lib().pxsom_comm_p2p_connect(self.handle, ctypes.cast(blob, ctypes.c_void_p),
                                                       P2P_HANDLE_BYTES * self.nranks), "pxsom_comm_p2p_connect")

    def allreduce_sum(self, t: torch.Tensor) -> None:
        if t.dtype != torch.float64 or not t.is_contiguous():
            raise ValueError("the exchange reduces contiguous float64 buffers")
        rc = _capi.lib().pxsom_comm_allreduce_sum_f64(self.handle, t.data_ptr(), t.numel(), _capi.stream_ptr())
        _capi.check(rc, "pxsom_comm_allreduce_sum_f64")

    def set_fused(self, on: bool) -> None:
        """The fused 10 x 10 training step runs the exchange inside its own launch (every rank: the same value)."""
        _capi.check(_capi.lib().pxsom_comm_p2p_set_fused(self.handle, 1 if on else 0), "pxsom_comm_p2p_set_fused")
        self.fused = bool(on)

    def error_epoch(self) -> int:
        """0, or the number of the first exchange a peer did not arrive at in time (its result was NaN)."""
        import ctypes
        out = ctypes.c_uint64(0)
        _capi.check(_capi.lib().pxsom_comm_p2p_error(self.handle, ctypes.byref(out)), "pxsom_comm_p2p_error")
        return int(out.value)

    def close(self) -> None:
        if self.handle is not None:
            h, self.handle = self.handle, None
            _capi.check(_capi.lib().pxsom_comm_destroy(h), "pxsom_comm_destroy")


def batch_train_finish(state: BatchTrainState, steps_done: int, total_steps: int, alpha_range, radius_range,
                       w_out: torch.Tensor) -> None:
    """Applies the last pending update of a run: ``w_out`` [K, C] receives the codebook after ``steps_done`` steps."""
    w_out = _codebook(w_out)
    sch = state.schedule
    rc = _capi.lib().pxsom_batch_train_sched_finish(
        state.wbuf.data_ptr(), state.ring.data_ptr(), state.xdim, state.ydim, state.c, sch.phases,
        state.edges.ctypes.data, sch.steps, int(steps_done), int(total_steps) // sch.steps, float(alpha_range[0]),
        float(alpha_range[1]), float(radius_range[0]), float(radius_range[1]), w_out.data_ptr(), _capi.stream_ptr())
    _capi.check(rc, "pxsom_batch_train_sched_finish")


ACC_PREPARED = 1  # include/pxsom.h PXSOM_ACC_PREPARED


def batch_accumulate(x: torch.Tensor, w: torch.Tensor, labels: torch.Tensor, stats: torch.Tensor,
                     workspace: AssignWorkspace, prepared: bool = False) -> None:
    """Zero ``stats`` ([K*C sums | K counts], float64), label every row of ``x`` and accumulate.
    ``prepared``: ``batch_update_prepare`` already cleared ``stats`` (and, for shapes the accumulating filter
    does not prepare itself, readied ``workspace`` for ``w``): no memset, no prep launch."""
    n, c, ldx, dt = _matrix_args(x)
    w = _codebook(w)
    k = w.shape[0]
    if not workspace.fits(n, c, k):
        raise ValueError("assign workspace too small for this mini-batch")
    if labels.dtype != torch.int32 or labels.numel() < n:
        raise ValueError("labels scratch must be int32 with at least n entries")
    if stats.dtype != torch.float64 or stats.numel() != k * (c + 1) or not stats.is_contiguous():
        raise ValueError("stats must be a contiguous float64 vector of K*(C+1) entries")
    rc = _capi.lib().pxsom_batch_accumulate(x.data_ptr(), n, c, ldx, dt, w.data_ptr(), k,
                                            labels.data_ptr(), stats.data_ptr(),
                                            workspace.buf.data_ptr(), workspace.bytes,
                                            ACC_PREPARED if prepared else 0, _capi.stream_ptr())
    _capi.check(rc, "pxsom_batch_accumulate")


def batch_update_prepare(w: torch.Tensor, xdim: int, ydim: int, stats: torch.Tensor, thr: float,
                         alpha: float, workspace: Optional[AssignWorkspace],
                         stats_next: Optional[torch.Tensor] = None) -> None:
    """Batch-rule codebook update from ``stats`` ([K*C sums | K counts], all-reduced), in place on ``w``.
    ``stats_next`` -- the buffer the next accumulate fills (alternate two) -- is cleared by the same launch
    (``None``: ``stats`` itself is cleared afterwards); ``workspace`` is prepared for the new codebook where
    the shape needs that.  Next accumulate: ``prepared=True``."""
    w = _codebook(w)
    k, c = w.shape
    if k != xdim * ydim:
        raise ValueError(f"codebook has {k} nodes, grid is {xdim}x{ydim}")
    if stats.dtype != torch.float64 or stats.numel() != k * (c + 1) or not stats.is_contiguous():
        raise ValueError("stats must be a contiguous float64 vector of K*(C+1) entries")
    if stats_next is not None and (stats_next.dtype != torch.float64 or stats_next.numel() != stats.numel()
                                   or not stats_next.is_contiguous()):
        raise ValueError("stats_next must look like stats")
    rc = _capi.lib().pxsom_batch_update_prepare(w.data_ptr(), int(xdim), int(ydim), c, stats.data_ptr(),
                                                stats_next.data_ptr() if stats_next is not None else None,
                                                float(thr), float(alpha),
                                                workspace.buf.data_ptr() if workspace is not None else None,
                                                workspace.bytes if workspace is not None else 0,
                                                _capi.stream_ptr())
    _capi.check(rc, "pxsom_batch_update_prepare")


ONLINE_INT_ABS = 1  # include/pxsom.h PXSOM_ONLINE_INT_ABS


def train_online(x: torch.Tensor, w: torch.Tensor, xdim: int, ydim: int, rlen: int,
                 alpha_range, radius_range, order: torch.Tensor, int_abs: bool = False) -> torch.Tensor:
    """Exact online SOM (FlowSOM C_SOM) in place on ``w`` [xdim*ydim, C] f64.  ``int_abs``: the other reading of the
    early-stop accumulator (``flowsom.RECALLED["change_abs"]``)."""
    n, c, ldx, dt = _matrix_args(x)
    w = _codebook(w)
    if w.shape != (xdim * ydim, c):
        raise ValueError(f"codebook shape {tuple(w.shape)} != ({xdim * ydim}, {c})")
    if order.dtype != torch.int64 or not order.is_cuda or order.numel() != n * rlen:
        raise ValueError("order must be an int64 HBM vector of n*rlen row indices")
    rc = _capi.lib().pxsom_train_online_ex(x.data_ptr(), n, c, ldx, dt, w.data_ptr(), int(xdim),
                                           int(ydim), int(rlen), float(alpha_range[0]),
                                           float(alpha_range[1]), float(radius_range[0]),
                                           float(radius_range[1]), order.data_ptr(),
                                           ONLINE_INT_ABS if int_abs else 0, _capi.stream_ptr())
    _capi.check(rc, "pxsom_train_online")
    return w


def batch_update(w: torch.Tensor, xdim: int, ydim: int, sums: torch.Tensor, counts: torch.Tensor,
                 thr: float, alpha: float) -> torch.Tensor:
    w = _codebook(w)
    c = w.shape[1]
    if sums.dtype != torch.float64 or counts.dtype != torch.float64:
        raise ValueError("sums and counts must be float64 (counts are exact integers)")
    rc = _capi.lib().pxsom_batch_update(w.data_ptr(), int(xdim), int(ydim), c, sums.data_ptr(),
                                        counts.data_ptr(), float(thr), float(alpha),
                                        _capi.stream_ptr())
    _capi.check(rc, "pxsom_batch_update")
    return w


def gaussian_kernel1d(sigma: float, truncate: float = 4.0):
    """scipy.ndimage._filters._gaussian_kernel1d(sigma, 0, radius): host numpy, as scipy builds it."""
    radius = int(truncate * float(sigma) + 0.5)
    sigma2 = sigma * sigma
    xs = np.arange(-radius, radius + 1)
    phi_x = np.exp(-0.5 / sigma2 * xs ** 2)
    phi_x = phi_x / phi_x.sum()
    return np.ascontiguousarray(phi_x[::-1], dtype=np.float64), radius


def gaussian_blur_hwc(img: torch.Tensor, sigma: float, tmp: Optional[torch.Tensor] = None,
                      f32_semantics: bool = False, generic_form: bool = False) -> torch.Tensor:
    """In-place per-channel Gaussian blur of an [H, W, C] float64 HBM image (scipy semantics).
    ``f32_semantics``: the values are widened float32 and every pass is stored as float32, as scipy does
    for a float32 image."""
    if img.dtype != torch.float64 or not img.is_cuda or not img.is_contiguous() or img.dim() != 3:
        raise ValueError("image must be a contiguous float64 [H, W, C] HBM tensor")
    h, w, c = img.shape
    if tmp is None:
        tmp = torch.empty_like(img)
    weights, radius = gaussian_kernel1d(sigma)
    rc = _capi.lib().pxsom_gaussian_blur_hwc(img.data_ptr(), tmp.data_ptr(), h, w, c,
                                             weights.ctypes.data, radius,
                                             int(bool(f32_semantics)) | (2 if generic_form else 0), _capi.stream_ptr())
    _capi.check(rc, "pxsom_gaussian_blur_hwc")
    return img


def rowsum_filter_normalize(x: torch.Tensor, thresh: float, f32_semantics: bool = False):
    """(rows [m, C] f64 = x_i / rowsum_i for kept pixels, flat pixel index [m] i64), compacted in order.
    ``f32_semantics``: widened float32 values, row sum and division in binary32 (a float32 pandas frame)."""
    if x.dtype != torch.float64 or not x.is_cuda or not x.is_contiguous() or x.dim() != 2:
        raise ValueError("matrix must be a contiguous float64 [N, C] HBM tensor")
    n, c = x.shape
    out = torch.empty_like(x)
    idx = torch.empty(n, dtype=torch.int64, device=x.device)
    cnt = torch.zeros(1, dtype=torch.int64, device=x.device)
    wsb = _capi.lib().pxsom_rownorm_workspace_bytes(n)
    ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=x.device)
    rc = _capi.lib().pxsom_rowsum_filter_normalize(x.data_ptr(), n, c, float(thresh), out.data_ptr(),
                                                   idx.data_ptr(), cnt.data_ptr(), ws.data_ptr(), wsb,
                                                   int(bool(f32_semantics)), _capi.stream_ptr())
    _capi.check(rc, "pxsom_rowsum_filter_normalize")
    m = int(cnt.item())
    return out[:m], idx[:m]


def normalize_columns(x: torch.Tensor, norm: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    if x.dtype != torch.float64 or norm.dtype != torch.float64:
        raise ValueError("normalize_columns works in float64 like the reference")
    n, c = x.shape
    if out is None:
        out = torch.empty((n, c), dtype=torch.float64, device=x.device)
    rc = _capi.lib().pxsom_normalize_columns(x.data_ptr(), n, c, x.stride(0) if n > 1 else c, norm.data_ptr(),
                                             out.data_ptr(), out.stride(0) if n > 1 else c, _capi.stream_ptr())
    _capi.check(rc, "pxsom_normalize_columns")
    return out


def quantile_nonzero(x: torch.Tensor, q: float, keep_mode: int = 0) -> torch.Tensor:
    """Exact type-7 quantile of the kept (non-zero / positive) values of every column -> [C] f64."""
    if x.dtype != torch.float64 or not x.is_cuda or x.dim() != 2 or x.stride(1) != 1:
        raise ValueError("matrix must be a float64 [N, C] HBM tensor with contiguous rows")
    n, c = x.shape
    out = torch.empty(c, dtype=torch.float64, device=x.device)
    wsb = _capi.lib().pxsom_quantile_workspace_bytes(n, c)
    ws = torch.empty(wsb, dtype=torch.uint8, device=x.device)
    rc = _capi.lib().pxsom_quantile_nonzero(x.data_ptr(), n, c, x.stride(0) if n > 1 else c, float(q),
                                            int(keep_mode), out.data_ptr(), ws.data_ptr(), wsb,
                                            _capi.stream_ptr())
    _capi.check(rc, "pxsom_quantile_nonzero")
    return out


def to_numpy(t: torch.Tensor) -> np.ndarray:
    return t.detach().cpu().numpy()


def pair_histogram(a: torch.Tensor, b: torch.Tensor, na: int, nb: int,
                   out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``out[a_i, b_i] += 1`` over two int32 HBM vectors (pairs outside [0, na) x [0, nb) are ignored);
    ``out`` [na, nb] int64 (zeros if not given)."""
    if a.dtype != torch.int32 or b.dtype != torch.int32 or a.numel() != b.numel() or not a.is_cuda:
        raise ValueError("a and b must be int32 HBM vectors of the same length")
    a, b = a.contiguous(), b.contiguous()
    if out is None:
        out = torch.zeros((int(na), int(nb)), dtype=torch.int64, device=a.device)
    rc = _capi.lib().pxsom_pair_histogram(a.data_ptr(), b.data_ptr(), a.numel(), int(na), int(nb),
                                          out.data_ptr(), _capi.stream_ptr())
    _capi.check(rc, "pxsom_pair_histogram")
    return out


def quantile_f32(x: torch.Tensor, q: float, keep_mode: int = 1) -> torch.Tensor:
    """``np.quantile`` of the kept values of every float32 column, in numpy's float32 arithmetic
    (keep_mode 1: > 0, 2: every non-NaN value) -> [C] float32."""
    if x.dtype != torch.float32 or not x.is_cuda or x.dim() != 2 or x.stride(1) != 1:
        raise ValueError("matrix must be a float32 [N, C] HBM tensor with contiguous rows")
    n, c = x.shape
    out = torch.empty(c, dtype=torch.float64, device=x.device)
    wsb = _capi.lib().pxsom_quantile_workspace_bytes(n, c)
    ws = torch.empty(wsb, dtype=torch.uint8, device=x.device)
    rc = _capi.lib().pxsom_quantile_f32(x.data_ptr(), n, c, x.stride(0) if n > 1 else c, float(q), int(keep_mode),
                                        out.data_ptr(), ws.data_ptr(), wsb, _capi.stream_ptr())
    _capi.check(rc, "pxsom_quantile_f32")
    return out.to(torch.float32)


def scaled_rowsum(img: torch.Tensor, norm: torch.Tensor) -> torch.Tensor:
    """``np.sum(img / norm, axis=-1)`` for [N, C] pixels and [C] divisors of one floating type (float32: numpy's
    binary32 arithmetic; float64: what numpy computes for every other image dtype), in numpy's summation
    order -> [N] of that type."""
    if img.dtype not in (torch.float32, torch.float64) or norm.dtype != img.dtype or not img.is_cuda \
            or img.dim() != 2 or img.stride(1) != 1:
        raise ValueError("img must be a float32 / float64 [N, C] HBM tensor with contiguous rows, norm [C] of the same type")
    n, c = img.shape
    out = torch.empty(n, dtype=img.dtype, device=img.device)
    fn = _capi.lib().pxsom_scaled_rowsum_f32 if img.dtype == torch.float32 else _capi.lib().pxsom_scaled_rowsum_f64
    rc = fn(img.data_ptr(), n, c, img.stride(0) if n > 1 else c, norm.contiguous().data_ptr(), out.data_ptr(),
            _capi.stream_ptr())
    _capi.check(rc, "pxsom_scaled_rowsum")
    return out


def scaled_rowsum_f32(img: torch.Tensor, norm: torch.Tensor) -> torch.Tensor:
    if img.dtype != torch.float32 or norm.dtype != torch.float32:
        raise ValueError("img must be a float32 [N, C] HBM tensor with contiguous rows, norm float32 [C]")
    return scaled_rowsum(img, norm)


MASK_BAD_LABEL, MASK_BAD_PIXEL = 1, 2   # include/pxsom.h PXSOM_MASK_*
LUT_UNMAPPED = -2 ** 31                 # PXSOM_LUT_UNMAPPED


def cluster_mask(row_index: torch.Tensor, column_index: torch.Tensor, labels: torch.Tensor, lut: torch.Tensor,
                 h: int, w: int):
    """``mask.ravel()[row_index * w + column_index] = lut[labels]`` on an int16 ``[h, w]`` image of zeros
    (last row wins for a pixel listed twice).  int64 HBM vectors, int32 LUT.  Returns ``(mask, status)``:
    status 0, or MASK_BAD_LABEL / MASK_BAD_PIXEL bits (synchronises to read it)."""
    for name, t in (("row_index", row_index), ("column_index", column_index), ("labels", labels)):
        if t.dtype != torch.int64 or not t.is_cuda or t.dim() != 1 or not t.is_contiguous():
            raise ValueError(f"{name} must be a contiguous int64 HBM vector")
    if not (row_index.numel() == column_index.numel() == labels.numel()):
        raise ValueError("row_index, column_index and labels must have one entry per table row")
    if lut.dtype != torch.int32 or not lut.is_cuda or not lut.is_contiguous():
        raise ValueError("lut must be a contiguous int32 HBM vector")
    dev = labels.device
    mask = torch.empty((int(h), int(w)), dtype=torch.int16, device=dev)
    status = torch.empty(1, dtype=torch.int32, device=dev)
    wsb = _capi.lib().pxsom_cluster_mask_workspace_bytes(int(h), int(w))
    ws = torch.empty(max(wsb, 8), dtype=torch.uint8, device=dev)
    rc = _capi.lib().pxsom_cluster_mask(row_index.data_ptr(), column_index.data_ptr(), labels.data_ptr(),
                                        labels.numel(), lut.data_ptr(), lut.numel(), int(h), int(w),
                                        mask.data_ptr(), status.data_ptr(), ws.data_ptr(), wsb, _capi.stream_ptr())
    _capi.check(rc, "pxsom_cluster_mask")
    return mask, int(status.item())


def relabel(labels: torch.Tensor, lut: torch.Tensor, fill: int = -1, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``out[i] = lut[labels[i]]`` (``fill`` for labels outside the table): SOM cluster -> meta cluster for labels
    that live in HBM.  int32 vectors; ``out`` may be ``labels``."""
    if labels.dtype != torch.int32 or lut.dtype != torch.int32 or not labels.is_cuda or not labels.is_contiguous():
        raise ValueError("labels and lut must be contiguous int32 HBM vectors")
    if out is None:
        out = torch.empty_like(labels)
    rc = _capi.lib().pxsom_relabel(labels.data_ptr(), labels.numel(), lut.contiguous().data_ptr(), lut.numel(), int(fill),
                                   out.data_ptr(), _capi.stream_ptr())
    _capi.check(rc, "pxsom_relabel")
    return out


class AssignSumsWorkspace:
    """Scratch for pxsom_assign_sums, reusable across calls of the same (n_max, c, k)."""

    def __init__(self, n_max: int, c: int, k: int, device):
        self.bytes = _capi.lib().pxsom_assign_sums_workspace_bytes(int(n_max), int(c), int(k))
        if self.bytes == 0:
            raise _capi.PxsomError(f"unsupported assign shape n={n_max} c={c} k={k}")
        self.n_max, self.c, self.k = int(n_max), int(c), int(k)
        # cleared ONCE: the library leaves the statistics region at its head zero after every successful call, and skips its own
        # clearing launch while ``clean`` says so (PXSOM_TABLES_SCRATCH_CLEAN); a failed call drops the promise
        self.buf = torch.zeros(self.bytes, dtype=torch.uint8, device=device)
        self.clean = True
        self.assign_offset = int(_capi.lib().pxsom_assign_sums_scratch_bytes(int(c), int(k)))   # where the assign workspace begins

    def fits(self, n: int, c: int, k: int) -> bool:
        return c == self.c and k == self.k and n <= self.n_max


TABLES_SCRATCH_CLEAN = 1  # include/pxsom.h PXSOM_TABLES_SCRATCH_CLEAN


def assign_sums(x: torch.Tensor, w: torch.Tensor, labels: Optional[torch.Tensor] = None,
                sums: Optional[torch.Tensor] = None, counts: Optional[torch.Tensor] = None,
                workspace: Optional[AssignSumsWorkspace] = None):
    """BMU labels of every row AND the per-label channel sums [K, C] f64 / counts [K] i64 (added into ``sums`` /
    ``counts``), reading ``x`` once where the shape allows.  Returns ``(labels, sums, counts)``."""
    n, c, ldx, dt = _matrix_args(x)
    w = _codebook(w)
    k = w.shape[0]
    if labels is None:
        labels = torch.empty(n, dtype=torch.int32, device=x.device)
    if sums is None:
        sums = torch.zeros((k, c), dtype=torch.float64, device=x.device)
    if counts is None:
        counts = torch.zeros(k, dtype=torch.int64, device=x.device)
    if workspace is None or not workspace.fits(n, c, k):
        workspace = AssignSumsWorkspace(n, c, k, x.device)
    flags, workspace.clean = (TABLES_SCRATCH_CLEAN if workspace.clean else 0), False
    rc = _capi.lib().pxsom_assign_sums_ex(x.data_ptr(), n, c, ldx, dt, w.data_ptr(), k, labels.data_ptr(), sums.data_ptr(),
                                          counts.data_ptr(), workspace.buf.data_ptr(), workspace.bytes, flags, _capi.stream_ptr())
    _capi.check(rc, "pxsom_assign_sums_ex")
    workspace.clean = True
    return labels, sums, counts


def assign_means(x: torch.Tensor, w: torch.Tensor, labels: torch.Tensor, sums: torch.Tensor, counts: torch.Tensor,
                 means: Optional[torch.Tensor], workspace: AssignSumsWorkspace) -> None:
    """Labels, per-cluster sums / counts (OVERWRITTEN) and means = sums / max(count, 1) in one library call and one pass
    over ``x`` where the shape allows (pxsom_assign_means)."""
    n, c, ldx, dt = _matrix_args(x)
    w = _codebook(w)
    k = w.shape[0]
    if not workspace.fits(n, c, k):
        raise ValueError("workspace does not fit this matrix")
    flags, workspace.clean = (TABLES_SCRATCH_CLEAN if workspace.clean else 0), False
    rc = _capi.lib().pxsom_assign_means_ex(x.data_ptr(), n, c, ldx, dt, w.data_ptr(), k, labels.data_ptr(), sums.data_ptr(),
                                           counts.data_ptr(), means.data_ptr() if means is not None else None,
                                           workspace.buf.data_ptr(), workspace.bytes, flags, _capi.stream_ptr())
    _capi.check(rc, "pxsom_assign_means_ex")
    workspace.clean = True
