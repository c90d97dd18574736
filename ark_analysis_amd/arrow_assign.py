"""Labelling a FOV table without going through pandas (SURVEY.md section 8 f, rank 1).

``PixelSOMCluster.assign_som_clusters`` is the reference-shaped entry point: DataFrame in, DataFrame out,
three full copies of the table on the host before a byte reaches the GPU.  ``cluster_pixels`` only needs
"feather file in, feather file with normalised channels + ``pixel_som_cluster`` out", so its hot loop works
on the Arrow table the file reader produced:

    column chunks --H2D--> [C, n] f64 --transpose--> [n, C] --pxsom_normalize_columns (IEEE division, the
    reference's ``x / norm``)--> pxsom_assign --> labels;   [n, C] --transpose--> [C, n] --D2H--> Arrow
    columns;  untouched columns (fov, row_index, ...) are passed through zero-copy.

The [C, n] host block the normalised channels land in is recycled (:class:`HostBlocks`): a device-to-host copy
into freshly allocated memory runs at the page-fault rate (measured: 32-40 ms per 185 MB table against 3.3 ms
into memory that has been touched before -- page-locked or not makes no difference on this platform, and
host-to-device from the reader's pageable buffers already runs at 3.5 ms), so ``cluster_pixels`` hands each
block back once the writer thread has serialised the table.

The result, read back with ``read_dataframe``, is identical (values, dtypes, column order, index) to what
the DataFrame path writes -- ``tests/test_pipeline_dropin.py`` holds the comparison -- and tables the fast
path does not cover (non-float64 channels, nulls, norm / codebook columns that differ) simply take the
DataFrame path.
"""
import json
import warnings
from typing import Optional

import numpy as np
import pandas as pd
import pyarrow as pa

LABEL_COLUMN = "pixel_som_cluster"


def _plain_f64(column: pa.ChunkedArray) -> bool:
    return column.type == pa.float64() and column.null_count == 0


def applicable(som, table: pa.Table, normalize: bool) -> bool:
    """Fast path: trained-on columns == norm columns (same order), all present as null-free float64."""
    if table.num_rows == 0 or som.weights is None:
        return False
    feats = list(som.weights.columns)
    if normalize and feats != list(som.norm_data.columns):
        return False
    names = set(table.column_names)
    return all(f in names and _plain_f64(table.column(f)) for f in feats)


def _label_field_metadata() -> dict:
    """The entry pandas writes for an int32 ``pixel_som_cluster`` column."""
    probe = pa.Table.from_pandas(pd.DataFrame({LABEL_COLUMN: np.zeros(1, dtype=np.int32)}), preserve_index=None)
    meta = json.loads(probe.schema.metadata[b"pandas"].decode())
    return next(c for c in meta["columns"] if c["name"] == LABEL_COLUMN)


def _with_label_metadata(schema_meta: Optional[dict], names) -> Optional[dict]:
    """Input table's pandas metadata with the column list brought in line with ``names``."""
    if not schema_meta or b"pandas" not in schema_meta:
        return schema_meta
    meta = json.loads(schema_meta[b"pandas"].decode())
    known = {c["name"]: c for c in meta.get("columns", [])}
    known[LABEL_COLUMN] = _label_field_metadata()
    index_entries = [c for c in meta.get("columns", []) if c["name"] not in names and c["name"] != LABEL_COLUMN]
    meta["columns"] = [known[n] for n in names if n in known] + index_entries
    out = dict(schema_meta)
    out[b"pandas"] = json.dumps(meta).encode()
    return out


class HostBlocks:
    """Recycled host blocks for the channel columns coming back from the GPU.  ``take(numel)`` hands out a
    float64 tensor of at least ``numel`` elements (a used one when available, the smallest that fits);
    ``give`` returns it (called from the writer threads).  The pool grows to the number of tables in flight.
    (A helper thread that touched blocks ahead of need was tried and made the caller's thread slower: its page
    zeroing competes with the copies for the same memory system.)"""

    def __init__(self):
        import threading
        self._free = []
        self._lock = threading.Lock()

    def take(self, numel: int):
        import torch
        with self._lock:
            fits = [i for i, b in enumerate(self._free) if b.numel() >= numel]
            if fits:
                return self._free.pop(min(fits, key=lambda i: self._free[i].numel()))
        return torch.empty(numel, dtype=torch.float64)

    def give(self, block) -> None:
        with self._lock:
            self._free.append(block)

    def close(self) -> None:
        with self._lock:
            self._free.clear()


def label_table(som, table: pa.Table, normalize: bool, blocks: Optional[HostBlocks] = None):
    """``table`` with the SOM's channels normalised (if ``normalize``) and ``pixel_som_cluster`` appended;
    ``som.som_clusters_seen`` is updated.  Requires :func:`applicable`.  With ``blocks`` the result is
    ``(table, release, totals)``: the channel columns live in a recycled host block and ``release()`` must be called
    when the table has been written (or dropped); ``totals`` = ``(channels, sums [K, C], counts [K])`` of the rows as
    written -- the per-cluster table ``generate_som_avg_files`` would otherwise re-read the file for."""
    import torch

    from . import _capi, som_device
    dev = _capi.require_gpu()
    feats = list(som.weights.columns)
    n, c = table.num_rows, len(feats)

    planar = torch.empty((c, n), dtype=torch.float64, device=dev)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", UserWarning)   # torch: "array is not writable" (we only read it)
        for j, name in enumerate(feats):
            at = 0
            for chunk in table.column(name).chunks:
                host = chunk.to_numpy(zero_copy_only=True)
                planar[j, at:at + len(host)].copy_(torch.from_numpy(host), non_blocking=True)
                at += len(host)
    rows = planar.t().contiguous()                       # [n, c] row-major, what the kernels read
    if normalize:
        norm = torch.from_numpy(som.norm_data.iloc[0].to_numpy(dtype=np.float64)).to(dev)
        som_device.normalize_columns(rows, norm, out=rows)   # element-wise, in place
    codebook = torch.from_numpy(np.ascontiguousarray(som.weights.to_numpy(dtype=np.float64))).to(dev)
    labels, _ = som_device.assign(rows, codebook)
    totals = None
    if blocks is not None:      # (the rows are in HBM now: their per-cluster sums cost 20 us, not a second file read)
        sums, counts = som_device.cluster_sums(rows, labels, codebook.shape[0])
        counts_host = counts.cpu().numpy()
        totals = (tuple(feats), sums.cpu().numpy(), counts_host)
        som.som_clusters_seen.update((np.flatnonzero(counts_host) + 1).tolist())
    else:
        som.som_clusters_seen.update(torch.unique(labels).cpu().tolist())
    label_array = pa.array(labels.cpu().numpy())         # int32, like the DataFrame path

    replaced, release = {}, None
    if normalize:
        planar = rows.t().contiguous()                   # [c, n]: one contiguous vector per channel
        if blocks is not None:
            block = blocks.take(c * n)
            back = block[:c * n].view(c, n)
            back.copy_(planar)                           # (synchronous: pageable destination)
            release = (lambda b=block: blocks.give(b))
            back = back.numpy()
        else:
            back = planar.cpu().numpy()
        replaced = {name: pa.array(back[j]) for j, name in enumerate(feats)}   # zero-copy views

    names, columns = [], []
    for name in table.column_names:
        if name == LABEL_COLUMN:
            continue                                      # re-labelling: the old column goes, the new one is appended
        names.append(name)
        columns.append(replaced.get(name, table.column(name)))
    names.append(LABEL_COLUMN)
    columns.append(label_array)
    out = pa.Table.from_arrays(columns, names=names)
    out = out.replace_schema_metadata(_with_label_metadata(table.schema.metadata, names))
    if blocks is None:
        return out
    return out, release, totals
