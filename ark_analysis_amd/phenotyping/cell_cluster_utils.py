"""``ark.phenotyping.cell_cluster_utils`` pieces next to the SOM path:
``compute_cell_som_cluster_cols_avg`` (/root/reference/src/ark/phenotyping/cell_cluster_utils.py:10-60;
a K-row reduction of the cell table, thousands of rows: stays on the host) and ``create_c2pc_data``
(:63-192; the cell x pixel-cluster count matrix that feeds the cell SOM -- its counting step is the device
histogram pxsom_pair_histogram)."""
import os
import warnings

import numpy as np
import pandas as pd

from .. import flowsom
from ..fov_tables import FovTableDir, unify_label_column
from ..host_utils import validate_paths, verify_in_list


def compute_cell_som_cluster_cols_avg(cell_cluster_data, cell_som_cluster_cols,
                                      cell_cluster_col, keep_count=False):
    """Mean of ``cell_som_cluster_cols`` per value of ``cell_cluster_col`` (cell SOM or meta cluster),
    one row per cluster in ascending order; ``keep_count`` appends the number of cells per cluster."""
    verify_in_list(provided_cluster_col=cell_cluster_col,
                   valid_cluster_cols=['cell_som_cluster', 'cell_meta_cluster'])
    verify_in_list(provided_cluster_col=cell_som_cluster_cols,
                   cluster_data_valid_cols=cell_cluster_data.columns.values)

    by_cluster = cell_cluster_data[list(cell_som_cluster_cols) + [cell_cluster_col]].groupby(cell_cluster_col)
    table = by_cluster.mean().reset_index()
    table[cell_cluster_col] = table[cell_cluster_col].astype(np.int64)
    if keep_count:
        table['count'] = by_cluster.size().to_numpy()
    return table


def _cluster_ids(values) -> np.ndarray:
    """The cluster column as the reference uses it (cell_cluster_utils.py:128-136): float columns become
    integers ("can happen with numeric types"), everything else -- integer ids, or the free-text names the
    meta-cluster remapping step assigns (``pixel_meta_cluster_rename``, e.g. 'CD4_T') -- is taken as it is."""
    arr = np.asarray(values)
    if arr.dtype.kind == 'f':
        return arr.astype(np.int64)
    return arr.astype(str) if arr.dtype.kind == 'O' else arr


def create_c2pc_data(fovs, pixel_data_path, cell_table_path,
                     pixel_cluster_col='pixel_meta_cluster_rename'):
    """How many pixels of every pixel cluster each cell contains.

    Returns ``(counts, counts / cell_size)``: one row per cell of ``fovs`` that is listed in the cell
    table *and* owns at least one clustered pixel; columns ``cell_size, fov, label`` followed by one
    ``<pixel_cluster_col>_<id>`` column per cluster met in the pixel tables (pandas' sorted-union order,
    i.e. as strings).  Clusters that no listed cell contains are dropped with a warning.

    The per-FOV ``groupby(['label', cluster]).size()`` + ``pivot`` of the reference is one device
    histogram (:func:`ark_analysis_amd.flowsom.pair_histogram`); the bookkeeping around it follows the
    reference's results: row order = cell-table order, count columns float64, zeros for absent pairs.

    One deliberate difference: the reference pairs its per-FOV count rows (taken in the iteration order of
    a Python ``set`` of labels, i.e. ascending for ordinary label ranges) with the cell-table rows (in
    table order) *by position* (:152-166), so a cell table that is not sorted by label within a FOV gets
    counts attached to the wrong cells there.  Here counts always go to the cell whose label they belong
    to; for label-sorted tables -- what ark's segmentation step writes, and what the golden fixture
    holds -- the two agree."""
    verify_in_list(provided_cluster_col=[pixel_cluster_col],
                   valid_cluster_cols=['pixel_som_cluster', 'pixel_meta_cluster_rename'])
    cells = pd.read_csv(cell_table_path)
    verify_in_list(required_cell_table_cols=['fov', 'label', 'cell_size'],
                   provided_cell_table_cols=cells.columns.values)
    cells = cells[['fov', 'label', 'cell_size']].copy()
    cells['label'] = cells['label'].astype(int)
    cells = cells[cells['fov'].isin(fovs)]

    tables = FovTableDir(pixel_data_path)
    per_fov = {}       # fov -> (cell-table row positions, cluster ids, counts [rows, clusters])
    met = set()        # every cluster id present in any FOV table
    for fov in fovs:
        pixels = unify_label_column(tables.load(fov))
        seg = pixels['label'].to_numpy().astype(np.int64)
        clu = _cluster_ids(pixels[pixel_cluster_col].to_numpy())
        ids, dense = np.unique(clu, return_inverse=True)
        ids = ids.tolist()                               # python ints / strs: dictionary keys and column names
        met.update(ids)
        hist = flowsom.pair_histogram(seg, dense, int(seg.max()) + 1 if seg.size else 1, len(ids))
        in_fov = np.flatnonzero((cells['fov'] == fov).to_numpy())
        cell_ids = cells['label'].to_numpy()[in_fov]
        # only cells that own at least one pixel take part (labels beyond the table's range own none)
        owns = (cell_ids >= 0) & (cell_ids < hist.shape[0])
        owns[owns] = hist[cell_ids[owns]].sum(axis=1) > 0
        per_fov[fov] = (in_fov[owns], ids, hist[cell_ids[owns]])

    order = sorted(met, key=str)                      # pandas unions the column labels as strings
    column_of = {cid: pos for pos, cid in enumerate(order)}
    counts = np.zeros((len(cells), len(order)), dtype=np.float64)
    for rows, ids, block in per_fov.values():
        counts[np.ix_(rows, [column_of[c] for c in ids])] = block

    count_cols = ['%s_%s' % (pixel_cluster_col, cid) for cid in order]
    out = pd.DataFrame(counts, columns=count_cols, index=cells.index)
    out.insert(0, 'label', cells['label'])
    out.insert(0, 'fov', cells['fov'])
    out.insert(0, 'cell_size', cells['cell_size'])
    out = out[out[count_cols].sum(axis=1) != 0].reset_index(drop=True)

    normed = out.copy()
    normed[count_cols] = normed[count_cols].div(normed['cell_size'], axis=0)

    empty = [c for c in count_cols if (normed[c] == 0).all()]
    if empty:
        warnings.warn('Pixel clusters %s do not appear in any cells, removed from analysis' % ','.join(empty))
        out, normed = out.drop(columns=empty), normed.drop(columns=empty)
    return out, normed


def add_consensus_labels_cell_table(base_dir, cell_table_path, cell_som_input_data):
    """The cell table with a ``cell_meta_cluster`` column (the renamed meta cluster of every cell; 'Unassigned' for
    cells the clustering never saw, e.g. too small to own a clustered pixel), saved beside it as
    ``<cell table>_cell_labels.csv`` (reference: cell_cluster_utils.py:195-247).  ``base_dir`` is unused, as there."""
    validate_paths([cell_table_path])
    cells = pd.read_csv(cell_table_path)
    if "segmentation_label" in cell_som_input_data.columns:      # (renamed in place, as the reference does)
        cell_som_input_data.rename(columns={"segmentation_label": "label"}, inplace=True)
    merged = cells.merge(cell_som_input_data, how="left", on=["fov", "label"])
    if "cell_size_y" in merged.columns.values:                   # both tables carry cell_size: keep the cell table's
        merged = merged.drop(columns=["cell_size_y"]).rename({"cell_size_x": "cell_size"}, axis=1)
    merged = merged[list(cells.columns.values) + ["cell_meta_cluster_rename"]]
    merged = merged.rename({"cell_meta_cluster_rename": "cell_meta_cluster"}, axis=1)
    merged["cell_meta_cluster"] = merged["cell_meta_cluster"].fillna("Unassigned")
    merged.to_csv(os.path.splitext(cell_table_path)[0] + "_cell_labels.csv", index=False)
