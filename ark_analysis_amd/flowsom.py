"""pyFlowSOM-compatible front end: ``som`` and ``map_data_to_nodes`` on MI355X.

These are the only two foreign functions the reference's hot path calls
(``from pyFlowSOM import map_data_to_nodes, som``,
/root/reference/src/ark/phenotyping/cluster_helpers.py:14; call sites :106-109 and :152-157).
Same names, argument meaning and return shapes, so ``cluster_helpers.py`` runs on the GPU by
changing that one import.

Seed handling (``som(..., seed=s)``): pyFlowSOM 0.1.16 is not available in this image, so its
random stream cannot be verified; by the reference's contract only *same seed => same weights*
holds (tests/phenotyping/cluster_helpers_test.py:323-332).  This build documents its own:
  initial nodes      data[RandomState(seed).choice(n, K, replace=False)]
  presentation order i_t = int(n * r_t / 2**31), r_t the glibc ``rand()`` stream after
                     ``srand(seed)`` (restated in C, pinned against libc in the tests)
Both are explicit inputs of the kernels (BASELINE.json: "given identical init weights and pixel
presentation order"), see :func:`som_with_inputs`.
"""
from typing import Optional, Sequence, Tuple

import numpy as np


def nhbrdist(xdim: int, ydim: int) -> np.ndarray:
    """Chebyshev distances between grid nodes, node k = x*ydim + y."""
    gx, gy = np.divmod(np.arange(xdim * ydim), ydim)
    return np.maximum(np.abs(gx[:, None] - gx[None, :]),
                      np.abs(gy[:, None] - gy[None, :])).astype(np.float64)


#: The details of pyFlowSOM.som this front end had to RECALL (the package is absent here: "parity unpinned"),
#: each with the alternatives a real vector dump (scripts/dump_pyflowsom_vectors.py) would discriminate between.
#: tests/test_pyflowsom_vectors.py sweeps them when tests/golden/g11_pyflowsom.npz exists.
RECALLED = {
    "radius_quantile": (0.67, "default radius_range = (quantile(nhbrdist, q), 0)"),
    "order_stream": ("glibc_rand", "presentation order: 'glibc_rand' i = int(n * rand()/2^31) after srand(seed); "
                                   "'numpy_randint' RandomState(seed) continued after the node choice; 'numpy_sample' "
                                   "int(n * random_sample()) of that stream"),
    "init_rule": ("numpy_choice", "initial nodes = data[RandomState(seed).choice(n, K, replace=False)]"),
    "node_order": ("xy", "node k = x*ydim + y ('yx': k = y*xdim + x)"),
    "change_abs": ("fabs", "the early-stop accumulator adds fabs(x - w) ('int_abs': C's integer abs(), which truncates "
                           "every |x - w| < 1 to 0 -- a run with rlen >= 2 on normalised data then stops at its second pass)"),
}


def default_radius_range(xdim: int, ydim: int, quantile: Optional[float] = None) -> Tuple[float, float]:
    """pyFlowSOM default: (quantile(nhbrdist, 0.67), 0) -- 6.0 for 10x10, 11.0 for 20x20."""
    q = RECALLED["radius_quantile"][0] if quantile is None else quantile
    return float(np.quantile(nhbrdist(xdim, ydim), q)), 0.0


def som_init_and_order(n: int, k: int, rlen: int, seed: Optional[int], order_stream: Optional[str] = None):
    """(indices of the K initial nodes, presentation order int64 [n*rlen]) for ``seed``.
    ``order_stream``: one of the alternatives listed in :data:`RECALLED` (default: the build's reading)."""
    from . import _capi
    if n < k:
        raise ValueError(f"som needs at least as many rows ({n}) as nodes ({k})")
    rs = np.random.RandomState(seed)
    init_idx = rs.choice(n, k, replace=False)
    stream = RECALLED["order_stream"][0] if order_stream is None else order_stream
    if stream == "numpy_randint":
        return init_idx, rs.randint(0, n, size=n * rlen).astype(np.int64)
    if stream == "numpy_sample":
        return init_idx, (n * rs.random_sample(n * rlen)).astype(np.int64)
    if stream != "glibc_rand":
        raise ValueError("unknown order_stream %r" % (stream,))
    stream_seed = int(seed) if seed is not None else int(rs.randint(1, 2 ** 31 - 1))
    r = _capi.glibc_rand(stream_seed, n * rlen)
    order = (n * (r.astype(np.float64) / 2147483648.0)).astype(np.int64)
    return init_idx, order


def _as_device_matrix(data, device):
    import torch
    if isinstance(data, torch.Tensor):
        return data.to(device)
    arr = np.asarray(data)
    if arr.dtype not in (np.float32, np.float64):
        arr = arr.astype(np.float64)
    return torch.from_numpy(np.ascontiguousarray(arr)).to(device)


def som_with_inputs(data, init_nodes, order, xdim: int = 10, ydim: int = 10, rlen: int = 10,
                    alpha_range: Sequence[float] = (0.05, 0.01),
                    radius_range: Optional[Sequence[float]] = None, change_abs: Optional[str] = None) -> np.ndarray:
    """Exact online SOM with explicit initial nodes [K, C] and presentation order [n*rlen].  ``change_abs``: one of the
    alternatives of :data:`RECALLED` (default: the build's reading)."""
    import torch
    from . import _capi, som_device
    dev = _capi.require_gpu()
    x = _as_device_matrix(data, dev)
    w = torch.from_numpy(np.ascontiguousarray(init_nodes, dtype=np.float64)).to(dev)
    od = torch.from_numpy(np.ascontiguousarray(order, dtype=np.int64)).to(dev)
    if radius_range is None:
        radius_range = default_radius_range(xdim, ydim)
    reading = RECALLED["change_abs"][0] if change_abs is None else change_abs
    if reading not in ("fabs", "int_abs"):
        raise ValueError("unknown change_abs %r" % (reading,))
    som_device.train_online(x, w, xdim, ydim, rlen, alpha_range, radius_range, od, int_abs=reading == "int_abs")
    return w.cpu().numpy()


def som(data, xdim: int = 10, ydim: int = 10, rlen: int = 10,
        alpha_range: Sequence[float] = (0.05, 0.01), radius_range=None, distf: int = 2,
        nodes=None, importance=None, seed=None) -> np.ndarray:
    """Drop-in for ``pyFlowSOM.som``: returns the trained codebook [xdim*ydim, C] (float64)."""
    if distf != 2:
        raise NotImplementedError("only the Euclidean distance (distf=2) is built; ark uses no other")
    data = np.asarray(data) if not hasattr(data, "is_cuda") else data
    if importance is not None:
        data = np.asarray(data, dtype=np.float64) * np.asarray(importance, dtype=np.float64)
    n = data.shape[0]
    k = xdim * ydim
    init_idx, order = som_init_and_order(n, k, rlen, seed)
    if nodes is None:
        src = data[init_idx]
        nodes = src.cpu().numpy() if hasattr(src, "cpu") else np.asarray(src)
    return som_with_inputs(data, np.asarray(nodes, dtype=np.float64), order, xdim, ydim, rlen,
                           alpha_range, radius_range)


def _batch_backend():
    """(device, kernel set) batch-mode training runs on: HBM and the HIP kernels.  (The CPU test suite swaps this
    for host memory and an oracle-backed stand-in to exercise the host logic; nothing else ever does.)"""
    from . import _capi
    from .distributed import HipKernels
    return _capi.require_gpu(), HipKernels()


def som_batch(data, xdim: int = 10, ydim: int = 10, rlen: int = 1,
              alpha_range: Sequence[float] = (0.05, 0.01), radius_range=None, nodes=None, seed=None,
              batch_steps=None) -> np.ndarray:
    """Throughput-mode SOM training (the batch rule of DESIGN.md "K6b"; oracle of record ``orc_som_batch_sched``):
    ``rlen`` passes over ``data`` -- THIS RANK's rows when a process group exists (the per-step statistics are
    all-reduced, every rank returns the same codebook), all rows otherwise -- on the schedule ``batch_steps`` names:
    an int (that many equal mini-batch steps per pass), ``None`` / "two-phase" (the default: 6 large steps while the
    neighbourhood radius is >= 1, 16 in the BMU-only tail, the last one larger) or a ``schedule.BatchSchedule``.
    Same arguments as :func:`som` otherwise.  Initial nodes: ``nodes`` if given, else
    ``rows[RandomState(seed).choice(n, K, replace=False)]`` -- of rank 0's rows when it holds at least K, else of the
    first K rows of every rank pooled in rank order (a cohort with enough rows never fails because rank 0's share is
    short).  Not the reference's algorithm: labels equal the batch oracle's, not those of an online pyFlowSOM run."""
    import torch
    from . import distributed
    dev, kernels = _batch_backend()
    rank, world = distributed.context()
    arr = data if isinstance(data, torch.Tensor) else np.asarray(data)
    n, c = int(arr.shape[0]), int(arr.shape[1])
    k = xdim * ydim

    def host_rows(sel):
        src = arr[sel]
        return src.cpu().numpy() if hasattr(src, "cpu") else np.asarray(src)

    if nodes is None:
        counts = distributed.allgather_objects(n)
        if sum(counts) < k:      # decided on the job's row count, the same way on every rank
            raise ValueError(f"som needs at least as many rows ({sum(counts)}) as nodes ({k})")
        if counts[0] >= k:
            nodes = distributed.on_rank0(lambda: host_rows(np.random.RandomState(seed).choice(n, k, replace=False)))
        else:
            pool = np.concatenate([p for p in distributed.allgather_objects(host_rows(slice(0, k))) if len(p)])
            nodes = pool[np.random.RandomState(seed).choice(len(pool), k, replace=False)]
    if radius_range is None:
        radius_range = default_radius_range(xdim, ydim)
    x = _as_device_matrix(arr, dev)
    w = torch.from_numpy(np.ascontiguousarray(nodes, dtype=np.float64).reshape(k, c)).to(dev)
    trainer = distributed.BatchSOMTrainer(xdim, ydim, c, dev, batch_steps=batch_steps, alpha_range=alpha_range,
                                          radius_range=radius_range, kernels=kernels)
    trainer.train(x, w, num_passes=rlen)
    return w.cpu().numpy()


def map_data_to_nodes(nodes, newdata, distf: int = 2):
    """Drop-in for ``pyFlowSOM.map_data_to_nodes``: (labels 1-based, distances) of every row."""
    import torch
    from . import _capi, som_device
    if distf != 2:
        raise NotImplementedError("only the Euclidean distance (distf=2) is built; ark uses no other")
    dev = _capi.require_gpu()
    x = _as_device_matrix(newdata, dev)
    if x.dim() == 1:
        x = x.reshape(1, -1)
    w = torch.from_numpy(np.ascontiguousarray(nodes, dtype=np.float64)).to(dev)
    labels, dists = som_device.assign(x, w, want_dists=True)
    return labels.cpu().numpy(), dists.cpu().numpy()


def cluster_sums(data, labels, k: int):
    """Per-label channel sums [k, C] (float64) and counts [k] (int64) of the rows of ``data``.

    Device-backed replacement for the pandas ``groupby(cluster)[channels].sum()`` / ``.size()``
    pair of compute_pixel_cluster_channel_avg (pixel_cluster_utils.py:369-384).  ``labels`` are
    1-based; rows with labels outside 1..k are ignored."""
    import torch
    from . import _capi, som_device
    dev = _capi.require_gpu()
    x = _as_device_matrix(data, dev)
    lab = torch.from_numpy(np.ascontiguousarray(labels, dtype=np.int32)).to(dev)
    sums, counts = som_device.cluster_sums(x, lab, int(k))
    return sums.cpu().numpy(), counts.cpu().numpy()


def pair_histogram(a, b, na: int, nb: int) -> np.ndarray:
    """[na, nb] int64 counts of the pairs (a_i, b_i) -- the device form of
    ``groupby([label, cluster]).size()`` + ``pivot`` (create_c2pc_data, cell_cluster_utils.py:128-141)."""
    import torch
    from . import _capi, som_device
    dev = _capi.require_gpu()
    ad = torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(dev)
    bd = torch.from_numpy(np.ascontiguousarray(b, dtype=np.int32)).to(dev)
    return som_device.pair_histogram(ad, bd, int(na), int(nb)).cpu().numpy()


class HostStaging:
    """One reusable host buffer for images on their way to the GPU: page-locked when a GPU is there (the upload
    then runs at the link's rate instead of through the driver's bounce buffers, ~3 GB/s), and never fresh memory
    after its first use (writing into fresh pages costs more than the arithmetic that fills them)."""

    def __init__(self):
        self._block = None

    def array(self, shape, dtype) -> np.ndarray:
        """A C-contiguous array of ``shape`` / ``dtype`` on the buffer (valid until the next call)."""
        import torch
        dtype = np.dtype(dtype)
        need = int(np.prod(shape)) * dtype.itemsize
        if self._block is None or self._block.numel() < need:
            self._block = None
            try:
                self._block = torch.empty(need, dtype=torch.uint8, pin_memory=torch.cuda.is_available())
            except RuntimeError:               # page-locking refused (limits): pageable memory still gets reused
                self._block = torch.empty(need, dtype=torch.uint8)
        return self._block[:need].numpy().view(dtype).reshape(shape)


def _image_to_device(image, dev, dtype=None):
    """Host image ``[H, W]`` / ``[H, W, C]`` -> contiguous device tensor of the same shape (converted to the
    torch ``dtype`` on the device when given).  A stack whose storage is channel-planar (what
    ``image_io.read_channels`` returns: a ``[C, H, W]`` buffer viewed as ``[H, W, C]``) is uploaded as it lies
    and interleaved on the device -- on the host that transpose costs more than decoding the TIFFs."""
    import torch
    arr = np.asarray(image)
    if arr.ndim == 3 and not arr.flags.c_contiguous and arr.transpose(2, 0, 1).flags.c_contiguous:
        t = torch.from_numpy(arr.transpose(2, 0, 1)).to(dev).permute(1, 2, 0)
    else:
        t = torch.from_numpy(np.ascontiguousarray(arr)).to(dev)
    if dtype is not None:
        t = t.to(dtype)
    return t.contiguous()


def positive_quantile_f32(image, q: float):
    """``np.quantile(plane[plane > 0], q)`` of every channel plane of an ``[H, W]`` or ``[H, W, C]`` image
    (NaN where nothing is positive): the per-channel percentile of calculate_channel_percentiles
    (pixel_cluster_utils.py:41-51).  Scalar for a single plane, ``[C]`` otherwise.  float32 images use
    numpy's float32 arithmetic, integer / float64 images binary64 (as numpy does)."""
    from . import _capi, som_device
    dev = _capi.require_gpu()
    image = np.asarray(image)
    columns = 1 if image.ndim <= 2 else image.shape[-1]
    if image.dtype == np.float32:
        got = som_device.quantile_f32(_image_to_device(image, dev).view(-1, columns), q, keep_mode=1)
    else:
        host = image if image.dtype == np.float64 else image.astype(np.float64)   # keeps a planar layout
        got = som_device.quantile_nonzero(_image_to_device(host, dev).view(-1, columns), q, keep_mode=1)
    got = got.cpu().numpy()
    return got[0] if image.ndim <= 2 else got


def total_intensity_quantile_f32(image_hwc, norm, q: float):
    """``np.quantile(np.sum(image / norm, axis=-1), q)`` for an [H, W, C] image and [C] divisors: the per-FOV
    percentile of calculate_pixel_intensity_percentile (pixel_cluster_utils.py:96-103).  float32 image AND
    float32 divisors (MIBI / MPLEX float exports): numpy's binary32 arithmetic throughout; anything else (uint16 /
    int16 exports, float64 divisors): numpy promotes the division to binary64 and so does this."""
    import torch
    from . import _capi, som_device
    dev = _capi.require_gpu()
    image_hwc = np.asarray(image_hwc)
    norm = np.asarray(norm)
    if image_hwc.dtype == np.float32 and norm.dtype == np.float32:
        pixels = _image_to_device(image_hwc, dev).view(-1, image_hwc.shape[-1])
        sums = som_device.scaled_rowsum(pixels, torch.from_numpy(np.ascontiguousarray(norm)).to(dev))
        return som_device.quantile_f32(sums.reshape(-1, 1), q, keep_mode=2).cpu().numpy()[0]
    pixels = _image_to_device(image_hwc, dev, torch.float64).view(-1, image_hwc.shape[-1])   # exact widening
    sums = som_device.scaled_rowsum(pixels, torch.from_numpy(np.ascontiguousarray(norm, dtype=np.float64)).to(dev))
    return som_device.quantile_nonzero(sums.reshape(-1, 1), q, keep_mode=2).cpu().numpy()[0]


def fov_pixel_rows(img_hwc, sigma: float, thresh: float, nonzero_q=None, blocks=None):
    """The numeric core of ``create_fov_pixel_data`` (pixie_preprocessing.py:45-75): per-channel Gaussian blur,
    keep pixels whose channel sum exceeds ``thresh`` and that are not all zero, divide the kept pixels by
    their channel sum.  Returns ``(rows [m, C], flat pixel index [m])``; a float32 image keeps the reference's
    float32 arithmetic (rows come back float32), anything else is processed in binary64.
    ``nonzero_q``: also return :func:`nonzero_quantiles` of the rows at that q, taken while they are still
    in HBM (a third element).  ``blocks`` (an ``arrow_assign.HostBlocks``): the rows land in a recycled host block
    instead of fresh memory (a device-to-host copy into fresh pages runs at the page-fault rate) and a last
    element ``release()`` hands the block back once the caller is done with the rows."""
    import torch
    from . import _capi, som_device
    dev = _capi.require_gpu()
    img_hwc = np.asarray(img_hwc)
    f32 = img_hwc.dtype == np.float32
    h, w, c = img_hwc.shape
    if img_hwc.dtype not in (np.float32, np.float64):
        img_hwc = img_hwc.astype(np.float64)
    img = _image_to_device(img_hwc, dev, torch.float64)     # the kernels work on binary64 storage
    som_device.gaussian_blur_hwc(img, float(sigma), f32_semantics=f32)
    rows, kept = som_device.rowsum_filter_normalize(img.view(h * w, c), float(thresh), f32_semantics=f32)
    tail = ()
    if blocks is None:
        values = (rows.to(torch.float32) if f32 else rows).cpu().numpy()
    else:
        out = rows.to(torch.float32) if f32 else rows
        block = blocks.take((out.numel() * out.element_size() + 7) // 8)
        host = block.view(out.dtype)[:out.numel()].view(out.shape)
        host.copy_(out)
        values, tail = host.numpy(), (lambda: blocks.give(block),)
    if nonzero_q is None:
        return (values, kept.cpu().numpy()) + tail
    quantiles = som_device.quantile_nonzero(rows, float(nonzero_q), keep_mode=0).cpu().numpy()
    return (values, kept.cpu().numpy(), quantiles) + tail


def nonzero_quantiles(matrix, q: float) -> np.ndarray:
    """Per column: linear-interpolation quantile of the non-zero, non-NaN entries in binary64 (NaN for a
    column without any) -- ``DataFrame.replace(0, nan).quantile(q)`` up to pandas' casts, which callers add."""
    import torch
    from . import _capi, som_device
    dev = _capi.require_gpu()
    m = torch.from_numpy(np.ascontiguousarray(matrix, dtype=np.float64)).to(dev)
    return som_device.quantile_nonzero(m, float(q), keep_mode=0).cpu().numpy()


def pixel_cluster_mask(row_index, column_index, labels, id_mapping: dict, shape) -> np.ndarray:
    """The relabel + scatter of ``generate_pixel_cluster_mask`` (utils/data_utils.py:532-553): an int16
    ``shape`` image of zeros with ``id_mapping[label]`` written at every listed pixel.  Raises ``KeyError``
    for a label the mapping lacks and ``IndexError`` for a pixel outside the image, as the reference's dict
    lookup and numpy assignment do."""
    import torch
    from . import _capi, som_device
    dev = _capi.require_gpu()
    labels = np.ascontiguousarray(labels, dtype=np.int64)
    keys = np.fromiter(id_mapping.keys(), dtype=np.int64, count=len(id_mapping))
    size = int(keys.max()) + 1 if keys.size and keys.max() >= 0 else 0
    lut = np.full(size, som_device.LUT_UNMAPPED, dtype=np.int64)
    for key, value in id_mapping.items():         # negative or fractional keys can never equal a label here
        if float(key).is_integer() and 0 <= int(key) < size:
            lut[int(key)] = int(value)
    # ids are narrowed to int16 the way numpy narrows them on assignment; only the marker stays out of range
    narrowed = np.where(lut == som_device.LUT_UNMAPPED, lut, lut.astype(np.int16)).astype(np.int32)

    def up(a, dtype):
        host = np.ascontiguousarray(a, dtype=dtype)
        return torch.from_numpy(host if host.flags.writeable else host.copy()).to(dev)   # Arrow columns are read-only
    mask, status = som_device.cluster_mask(up(row_index, np.int64), up(column_index, np.int64), up(labels, np.int64),
                                           up(narrowed, np.int32), int(shape[0]), int(shape[1]))
    if status & som_device.MASK_BAD_PIXEL:
        raise IndexError("pixel coordinates outside the %d x %d image" % (shape[0], shape[1]))
    if status & som_device.MASK_BAD_LABEL:
        missing = sorted(set(np.unique(labels).tolist()) - set(int(k) for k in id_mapping))
        raise KeyError(missing[0] if missing else "cluster label without a cluster_id")
    return mask.cpu().numpy()
