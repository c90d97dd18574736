"""Labelling a FOV table without going through pandas (SURVEY.md section 8 f, rank 1).

``PixelSOMCluster.assign_som_clusters`` is the reference-shaped entry point: DataFrame in, DataFrame out,
three full copies of the table on the host before a byte reaches the GPU.  ``cluster_pixels`` only needs
"feather file in, feather file with normalised channels + ``pixel_som_cluster`` out", so its hot loop works
on the Arrow table the file reader produced:

    column chunks --H2D--> [C, n] f64 --transpose--> [n, C] --pxsom_normalize_columns (IEEE division, the
    reference's ``x / norm``)--> pxsom_assign --> labels;   [n, C] --transpose--> [C, n] --D2H--> Arrow
    columns;  untouched columns (fov, row_index, ...) are passed through zero-copy.

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


def label_table(som, table: pa.Table, normalize: bool) -> pa.Table:
    """``table`` with the SOM's channels normalised (if ``normalize``) and ``pixel_som_cluster`` appended;
    ``som.som_clusters_seen`` is updated.  Requires :func:`applicable`."""
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
    som.som_clusters_seen.update(torch.unique(labels).cpu().tolist())
    label_array = pa.array(labels.cpu().numpy())         # int32, like the DataFrame path

    replaced = {}
    if normalize:
        back = rows.t().contiguous().cpu().numpy()       # [c, n]: one contiguous vector per channel
        replaced = {name: pa.array(back[j]) for j, name in enumerate(feats)}

    names, columns = [], []
    for name in table.column_names:
        if name == LABEL_COLUMN:
            continue                                      # re-labelling: the old column goes, the new one is appended
        names.append(name)
        columns.append(replaced.get(name, table.column(name)))
    names.append(LABEL_COLUMN)
    columns.append(label_array)
    out = pa.Table.from_arrays(columns, names=names)
    return out.replace_schema_metadata(_with_label_metadata(table.schema.metadata, names))
