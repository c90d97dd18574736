"""``ark.phenotyping.cell_meta_clustering`` (/root/reference/src/ark/phenotyping/cell_meta_clustering.py:10-330): the
meta-clustering of the cell SOM clusters.  Everything here is host work on small tables -- one row per cell SOM cluster
for the consensus step (``cluster_helpers.PixieConsensusCluster``, Ward on the K x C average table), one row per cell
for the label lookups -- and exists so that the cell half of the Pixie workflow runs to its end on this package; the
device work of the cell path is the SOM itself (``cell_som_clustering``).  Signatures, messages and files as in the
reference; pinned by ``tests/golden/g12_cell_meta_clustering.npz`` (the reference's own run)."""
import os

import numpy as np
import pandas as pd

from ..host_utils import validate_paths, verify_in_list
from . import cell_cluster_utils, cluster_helpers

_SOM, _META, _NAME = "cell_som_cluster", "cell_meta_cluster", "cell_meta_cluster_rename"


def cell_consensus_cluster(base_dir, cell_som_cluster_cols, cell_som_input_data,
                           cell_som_expr_col_avg_name, max_k=20, cap=3, seed=42, overwrite=False):
    """Consensus (Ward) clustering of the cell SOM clusters' average table into ``max_k`` meta clusters and the meta
    label of every cell.  Returns ``(PixieConsensusCluster, cell_som_input_data with cell_meta_cluster)``; with meta
    labels already present and no ``overwrite`` the data come back untouched (the consensus object unfitted)."""
    avg_path = os.path.join(base_dir, cell_som_expr_col_avg_name)
    validate_paths([avg_path])
    verify_in_list(provided_cluster_cols=cell_som_cluster_cols,
                   som_cluster_counts_cols=pd.read_csv(avg_path, nrows=1).columns.values)
    cell_cc = cluster_helpers.PixieConsensusCluster("cell", avg_path, cell_som_cluster_cols, max_k=max_k, cap=cap)

    if _META in cell_som_input_data:
        if not overwrite:
            print("Meta clusters already assigned to each cell")
            return cell_cc, cell_som_input_data
        print("Overwrite flag set, reassigning meta cluster labels")
        cell_som_input_data = cell_som_input_data.drop(columns=_META)

    print("z-score scaling and capping data")
    cell_cc.scale_data()
    np.random.seed(seed)
    print("Running consensus clustering")
    cell_cc.run_consensus_clustering()
    print("Mapping cell data to consensus cluster labels")
    cell_cc.generate_som_to_meta_map()
    return cell_cc, cell_cc.assign_consensus_labels(cell_som_input_data)


def generate_meta_avg_files(base_dir, cell_cc, cell_som_cluster_cols, cell_som_input_data,
                            cell_som_expr_col_avg_name, cell_meta_expr_col_avg_name, overwrite=False):
    """Writes the per-meta-cluster average table (with counts) and adds the meta cluster column to the per-SOM-cluster
    average table on disk."""
    som_avg_path = os.path.join(base_dir, cell_som_expr_col_avg_name)
    meta_avg_path = os.path.join(base_dir, cell_meta_expr_col_avg_name)
    validate_paths([som_avg_path])
    if _META not in cell_som_input_data.columns.values:
        raise ValueError("cell_som_input_data does not have meta labels assigned")
    if os.path.exists(meta_avg_path):
        if not overwrite:
            print("Already generated average expression file for cell meta clusters, skipping")
            return
        print("Overwrite flag set, regenerating average expression file for cell meta clusters")

    print("Computing the average value of each training column specified per cell meta cluster")
    cell_cluster_utils.compute_cell_som_cluster_cols_avg(
        cell_som_input_data, cell_som_cluster_cols, _META, keep_count=True).to_csv(meta_avg_path, index=False)

    print("Mapping meta cluster values onto average expression values across cell SOM clusters")
    som_avg = pd.read_csv(som_avg_path)
    som_avg[_SOM] = som_avg[_SOM].astype(int)
    som_avg = som_avg.drop(columns=_META, errors="ignore")
    # (an as-of merge on the sorted SOM cluster id, as the reference: every id present in the mapping meets itself)
    pd.merge_asof(som_avg, cell_cc.mapping, on=_SOM).to_csv(som_avg_path, index=False)


def _remapping(base_dir, cell_remapped_name):
    """The user's remapping file -> (table, {som cluster: meta cluster}, {meta cluster: its name})."""
    path = os.path.join(base_dir, cell_remapped_name)
    table = pd.read_csv(path)
    verify_in_list(required_cols=[_SOM, _META, _NAME], remapped_data_cols=table.columns.values)
    to_meta = dict(table[[_SOM, _META]].values)
    to_name = dict(table[[_META, _NAME]].drop_duplicates().values)
    return table, to_meta, to_name


def apply_cell_meta_cluster_remapping(base_dir, cell_som_input_data, cell_remapped_name):
    """Relabels ``cell_meta_cluster`` by the remapping file and adds ``cell_meta_cluster_rename`` (in place, and
    returned)."""
    validate_paths([os.path.join(base_dir, cell_remapped_name)])
    table, to_meta, to_name = _remapping(base_dir, cell_remapped_name)
    cluster_helpers.verify_unique_meta_clusters(table, meta_cluster_type="cell")
    print("Using re-mapping scheme to re-label cell meta clusters")
    verify_in_list(fov_som_labels=cell_som_input_data[_SOM], som_labels_in_mapping=list(to_meta.keys()))
    cell_som_input_data[_META] = cell_som_input_data[_SOM].map(to_meta)
    cell_som_input_data[_NAME] = cell_som_input_data[_META].map(to_name)
    return cell_som_input_data


def generate_remap_avg_count_files(base_dir, cell_som_input_data, cell_remapped_name, cell_som_cluster_cols,
                                   cell_som_expr_col_avg_name, cell_meta_expr_col_avg_name):
    """Rewrites both average tables after a remapping: per-meta-cluster averages recomputed (with the names), the
    per-SOM-cluster table relabelled."""
    som_avg_path = os.path.join(base_dir, cell_som_expr_col_avg_name)
    meta_avg_path = os.path.join(base_dir, cell_meta_expr_col_avg_name)
    validate_paths([os.path.join(base_dir, cell_remapped_name), som_avg_path, meta_avg_path])
    _, to_meta, to_name = _remapping(base_dir, cell_remapped_name)

    print("Re-compute average value of each training column specified per cell meta cluster")
    meta_avg = cell_cluster_utils.compute_cell_som_cluster_cols_avg(cell_som_input_data, cell_som_cluster_cols, _META,
                                                                    keep_count=True)
    meta_avg[_NAME] = meta_avg[_META].map(to_name)
    meta_avg.to_csv(meta_avg_path, index=False)

    print("Re-assigning meta cluster column in cell SOM cluster average pixel cluster counts data")
    som_avg = pd.read_csv(som_avg_path)
    som_avg[_META] = som_avg[_SOM].map(to_meta)
    som_avg[_NAME] = som_avg[_META].map(to_name)
    som_avg.to_csv(som_avg_path, index=False)
