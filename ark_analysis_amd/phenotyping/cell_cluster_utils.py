"""``compute_cell_som_cluster_cols_avg`` of ``ark.phenotyping.cell_cluster_utils``
(/root/reference/src/ark/phenotyping/cell_cluster_utils.py:10-60): a K-row pandas groupby on the
cell table (thousands of rows -- stays on the host)."""
import numpy as np

from ..host_utils import verify_in_list


def compute_cell_som_cluster_cols_avg(cell_cluster_data, cell_som_cluster_cols,
                                      cell_cluster_col, keep_count=False):
    """Average of ``cell_som_cluster_cols`` per cell SOM / meta cluster."""
    verify_in_list(
        provided_cluster_col=cell_cluster_col,
        valid_cluster_cols=['cell_som_cluster', 'cell_meta_cluster']
    )
    verify_in_list(
        provided_cluster_col=cell_som_cluster_cols,
        cluster_data_valid_cols=cell_cluster_data.columns.values
    )

    cell_cluster_data_subset = cell_cluster_data.loc[
        :, list(cell_som_cluster_cols) + [cell_cluster_col]
    ]

    mean_count_totals = cell_cluster_data_subset.groupby(cell_cluster_col).mean().reset_index()
    mean_count_totals[cell_cluster_col] = mean_count_totals[cell_cluster_col].astype(np.int64)

    if keep_count:
        cell_cluster_totals = cell_cluster_data_subset.groupby(
            cell_cluster_col
        ).size().to_frame('count')
        cell_cluster_totals = cell_cluster_totals.reset_index(drop=True)
        mean_count_totals['count'] = cell_cluster_totals['count']
    return mean_count_totals
