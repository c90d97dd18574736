"""``ark.phenotyping.cell_cluster_utils`` pieces next to the SOM path:
``compute_cell_som_cluster_cols_avg`` (/root/reference/src/ark/phenotyping/cell_cluster_utils.py:10-60;
a K-row reduction of the cell table, thousands of rows: stays on the host) and ``create_c2pc_data``
(:63-192; the cell x pixel-cluster count matrix that feeds the cell SOM -- its counting step is the device
histogram pxsom_pair_histogram)."""
import numpy as np

from ..host_utils import verify_in_list


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
