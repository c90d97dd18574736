"""Cell SOM pipeline -- ``train_cell_som`` / ``cluster_cells`` / ``generate_som_avg_files`` as
``ark.phenotyping.cell_som_clustering`` offers them
(/root/reference/src/ark/phenotyping/cell_som_clustering.py:8-75, :78-139, :142-191): same arguments,
checks and printed lines.  The cell x feature table lives in memory (BASELINE.json config 4: 1e6 cells x
100 pixel-cluster counts); training and assignment are the same gfx950 kernels the pixel SOM uses.
"""
import os

from ..host_utils import validate_paths, verify_in_list
from . import cell_cluster_utils, cluster_helpers

#: columns of the cell table that are never SOM features
_NOT_FEATURES = ('fov', 'label', 'cell_size')


def train_cell_som(fovs, base_dir, cell_table_path, cell_som_cluster_cols,
                   cell_som_input_data, som_weights_name='cell_som_weights.feather',
                   xdim=10, ydim=10, lr_start=0.05, lr_end=0.01, num_passes=1, seed=42,
                   overwrite=False, normalize=True, *, train_mode="online", batch_steps=None):
    """Train the cell SOM on ``cell_som_cluster_cols`` of ``cell_som_input_data`` (rows of ``fovs``) and
    store the codebook in ``base_dir/som_weights_name``; returns the ``CellSOMCluster``.
    ``train_mode`` / ``batch_steps`` (keyword-only, beyond the reference): see ``train_pixel_som``."""
    from .. import distributed
    distributed.init_from_env()
    validate_paths([cell_table_path])
    verify_in_list(provided_cluster_cols=cell_som_cluster_cols,
                   som_input_cluster_cols=cell_som_input_data.columns.values)

    som = cluster_helpers.CellSOMCluster(
        cell_som_input_data, os.path.join(base_dir, som_weights_name), fovs, cell_som_cluster_cols,
        num_passes=num_passes, xdim=xdim, ydim=ydim, lr_start=lr_start, lr_end=lr_end, seed=seed,
        normalize=normalize, train_mode=train_mode, batch_steps=batch_steps)
    if distributed.context()[0] == 0:
        print("Training SOM")
    som.train_som(overwrite=overwrite)
    return som


def cluster_cells(base_dir, cell_pysom, cell_som_cluster_cols, num_parallel_cells=1000000,
                  overwrite=False):
    """Label every cell of ``cell_pysom.cell_data`` with its SOM cluster (``cell_som_cluster``) and
    return the table.  Labels from an earlier call are kept unless ``overwrite``."""
    if cell_pysom.weights is None:
        raise ValueError("Using untrained cell_pysom object, please invoke train_cell_som first")

    cells = cell_pysom.cell_data
    if "segmentation_label" in cells.columns:
        cells.rename(columns={"segmentation_label": "label"}, inplace=True)

    labelled_before = 'cell_som_cluster' in cells.columns.values
    if labelled_before and not overwrite:
        print("SOM clusters already assigned to each cell")
        return cells
    if labelled_before:
        print("Overwrite flag set, reassigning SOM cluster labels")

    # what is left after the bookkeeping columns must cover the columns the codebook was trained on
    # (cell_size is only present for pixel-cluster-count inputs)
    aside = [c for c in _NOT_FEATURES + ('cell_som_cluster',)
             if c in cells.columns.values or c in ('fov', 'label')]
    verify_in_list(cell_weights_columns=cell_pysom.weights.columns.values,
                   cell_som_input_data_columns=cells.drop(columns=aside).columns.values)

    print("Mapping cell data to SOM cluster labels")
    return cell_pysom.assign_som_clusters(num_parallel_cells)


def generate_som_avg_files(base_dir, cell_som_input_data, cell_som_cluster_cols,
                           cell_som_expr_col_avg_name, overwrite=False):
    """Write the mean of every training column per cell SOM cluster (plus cell counts) as CSV to
    ``base_dir/cell_som_expr_col_avg_name``."""
    target = os.path.join(base_dir, cell_som_expr_col_avg_name)
    if 'cell_som_cluster' not in cell_som_input_data.columns.values:
        raise ValueError('cell_som_input_data does not have SOM labels assigned')

    if os.path.exists(target):
        if not overwrite:
            print("Already generated average expression file for each cell SOM column, skipping")
            return
        print("Overwrite flag set, regenerating average expression file for cell SOM clusters")

    print("Computing the average value of each training column specified per cell SOM cluster")
    cell_cluster_utils.compute_cell_som_cluster_cols_avg(
        cell_som_input_data, cell_som_cluster_cols, 'cell_som_cluster', keep_count=True
    ).to_csv(target, index=False)
