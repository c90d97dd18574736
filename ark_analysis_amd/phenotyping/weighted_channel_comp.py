"""``ark.phenotyping.weighted_channel_comp`` without its plotting function
(/root/reference/src/ark/phenotyping/weighted_channel_comp.py:14-412): the channel expression of every cell as the
pixel-cluster-count-weighted mix of the pixel clusters' average expression, and its averages per cell SOM / meta
cluster.  Host work: one ``[cells, clusters] x [clusters, channels]`` product (tens of megaflops per FOV) and pandas
group-bys over one row per cell; only used to annotate the cell clusters for display.  Pinned by the reference's own
run (``tests/golden/g13_weighted_channel.npz``)."""
import os

import numpy as np
import pandas as pd

from ..fov_tables import read_dataframe
from ..host_utils import validate_paths, verify_in_list, verify_same_elements

_SOM, _META, _NAME = "cell_som_cluster", "cell_meta_cluster", "cell_meta_cluster_rename"


def compute_p2c_weighted_channel_avg(pixel_channel_avg, channels, cell_counts, fovs=None,
                                     pixel_cluster_col='pixel_meta_cluster_rename'):
    """Per cell: ``sum_k count[cell, k] * mean_expression[k, channel] / cell_size``.  ``cell_counts`` carries one
    ``<pixel_cluster_col>_<id>`` column per pixel cluster (create_c2pc_data's table); returns
    ``channels + [cell_size, fov, label]``, one row per cell of ``fovs``, in ``cell_counts`` order."""
    if "segmentation_label" in cell_counts.columns:          # (renamed in place, as the reference does)
        cell_counts.rename(columns={"segmentation_label": "label"}, inplace=True)
    if fovs is None:
        fovs = list(cell_counts['fov'].unique())
    else:
        verify_in_list(provided_fovs=fovs, dataset_fovs=cell_counts['fov'].unique())
    verify_in_list(provided_cluster_col=pixel_cluster_col,
                   valid_cluster_cols=['pixel_som_cluster', 'pixel_meta_cluster_rename'])

    cells = cell_counts[cell_counts['fov'].isin(fovs)]
    count_cols = sorted(c for c in cells.columns.values if pixel_cluster_col in c)
    ids = [c.replace(pixel_cluster_col + '_', '') for c in count_cols]

    # cluster ids are compared as strings (they come out of column names); rows follow the sorted id order
    if np.issubdtype(pixel_channel_avg[pixel_cluster_col].dtype, np.integer):
        pixel_channel_avg[pixel_cluster_col] = pixel_channel_avg[pixel_cluster_col].astype(str)
    means = pixel_channel_avg.sort_values(by=pixel_cluster_col)
    means = means[means[pixel_cluster_col].isin(ids)]
    verify_same_elements(enforce_order=True, cell_counts_cluster_ids=ids,
                         pixel_channel_cluster_ids=means[pixel_cluster_col].values)
    verify_in_list(provided_channels=channels, pixel_channel_avg_cols=means.columns.values)

    out = pd.DataFrame(np.matmul(cells[count_cols].values, means[channels].values), columns=channels)
    carried = cells.reset_index(drop=True)[['cell_size', 'fov', 'label']]
    out[['cell_size', 'fov', 'label']] = carried
    out[channels] = out[channels].div(out['cell_size'], axis=0)
    return out


def compute_cell_cluster_weighted_channel_avg(fovs, channels, base_dir, weighted_cell_channel_name,
                                              cell_cluster_data, cell_cluster_col='cell_meta_cluster'):
    """Mean weighted channel expression per cell SOM / meta cluster over the cells of ``fovs``."""
    path = os.path.join(base_dir, weighted_cell_channel_name)
    validate_paths([path])
    verify_in_list(provided_cluster_col=[cell_cluster_col], valid_cluster_cols=[_SOM, _META])

    table = read_dataframe(path)
    table = table[table['fov'].isin(fovs)].sort_values(by=['fov', 'label']).reset_index(drop=True)
    clusters = cell_cluster_data.sort_values(by=['fov', 'label']).reset_index(drop=True)
    verify_same_elements(enforce_order=True, cell_table_fovs=list(table['fov']), cluster_data_fovs=list(clusters['fov']))
    verify_same_elements(enforce_order=True, cell_table_labels=list(table['label']),
                         cluster_data_labels=list(clusters['label']))
    table[cell_cluster_col] = clusters[cell_cluster_col]
    means = table[list(channels) + [cell_cluster_col]].groupby(cell_cluster_col).mean().reset_index()
    means[cell_cluster_col] = means[cell_cluster_col].astype(dtype=int)
    return means


def generate_wc_avg_files(fovs, channels, base_dir, cell_cc, cell_som_input_data,
                          weighted_cell_channel_name='weighted_cell_channel.feather',
                          cell_som_cluster_channel_avg_name='cell_som_cluster_channel_avg.csv',
                          cell_meta_cluster_channel_avg_name='cell_meta_cluster_channel_avg.csv',
                          overwrite=False):
    """Writes the weighted channel averages per cell SOM cluster (with its meta cluster) and per cell meta cluster."""
    som_path = os.path.join(base_dir, cell_som_cluster_channel_avg_name)
    meta_path = os.path.join(base_dir, cell_meta_cluster_channel_avg_name)
    validate_paths([os.path.join(base_dir, weighted_cell_channel_name)])
    if os.path.exists(som_path) and os.path.exists(meta_path):
        if not overwrite:
            print("Already generated average weighted channel expression files, skipping")
            return
        print("Overwrite flag set, regenerating average weighted channel expression files")

    print("Compute average weighted channel expression across cell SOM clusters")
    som_avg = compute_cell_cluster_weighted_channel_avg(fovs, channels, base_dir, weighted_cell_channel_name,
                                                        cell_som_input_data, _SOM)
    print("Mapping meta cluster values onto average weighted channel expression"
          "across cell SOM clusters")
    pd.merge_asof(som_avg, cell_cc.mapping, on=_SOM).to_csv(som_path, index=False)

    print("Compute average weighted channel expression across cell meta clusters")
    compute_cell_cluster_weighted_channel_avg(fovs, channels, base_dir, weighted_cell_channel_name,
                                              cell_som_input_data, _META).to_csv(meta_path, index=False)


def generate_remap_avg_wc_files(fovs, channels, base_dir, cell_som_input_data, cell_remapped_name,
                                weighted_cell_channel_name, cell_som_cluster_channel_avg_name,
                                cell_meta_cluster_channel_avg_name):
    """Rewrites both weighted-channel average files after a manual remapping of the meta clusters."""
    remap_path = os.path.join(base_dir, cell_remapped_name)
    som_path = os.path.join(base_dir, cell_som_cluster_channel_avg_name)
    meta_path = os.path.join(base_dir, cell_meta_cluster_channel_avg_name)
    validate_paths([remap_path, os.path.join(base_dir, weighted_cell_channel_name), som_path, meta_path])
    remap = pd.read_csv(remap_path)
    verify_in_list(required_cols=[_SOM, _META, _NAME], remapped_data_cols=remap.columns.values)
    to_meta = dict(remap[[_SOM, _META]].values)
    to_name = dict(remap[[_META, _NAME]].drop_duplicates().values)

    print("Re-compute average weighted channel expression across cell meta clusters")
    meta_avg = compute_cell_cluster_weighted_channel_avg(fovs, channels, base_dir, weighted_cell_channel_name,
                                                         cell_som_input_data, _META)
    meta_avg[_NAME] = meta_avg[_META].map(to_name)
    meta_avg.to_csv(meta_path, index=False)

    print("Re-assigning meta cluster column in cell SOM cluster average weighted channel data")
    som_avg = pd.read_csv(som_path)
    som_avg[_META] = som_avg[_SOM].map(to_meta)
    som_avg[_NAME] = som_avg[_META].map(to_name)
    som_avg.to_csv(som_path, index=False)
