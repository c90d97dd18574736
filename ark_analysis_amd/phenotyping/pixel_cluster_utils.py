"""Hot-path pieces of ``ark.phenotyping.pixel_cluster_utils``
(/root/reference/src/ark/phenotyping/pixel_cluster_utils.py): row normalisation, the
per-cluster channel-average table and the restart helper.  The per-cluster reduction runs on
the GPU (pxsom_cluster_sums); the TIFF-bound helpers of the reference module are out of scope
(SURVEY.md section 2, row 5)."""
import os
import random
import warnings

import numpy as np
import pandas as pd
from pyarrow.lib import ArrowInvalid

from .. import flowsom
from ..host_utils import list_files, remove_file_extensions, validate_paths, verify_in_list
from .cluster_helpers import read_dataframe


def normalize_rows(pixel_data, channels, include_seg_label=True):
    """Divide each row by its channel sum (reference: pixel_cluster_utils.py:109-142)."""
    pixel_data_sub = pixel_data[channels]
    pixel_data_sub = pixel_data_sub.div(pixel_data_sub.sum(axis=1), axis=0)

    meta_cols = ['fov', 'row_index', 'column_index']
    if include_seg_label:
        meta_cols.append('label')

    pixel_data_sub[meta_cols] = pixel_data.loc[pixel_data_sub.index.values, meta_cols]
    return pixel_data_sub


def compute_pixel_cluster_channel_avg(fovs, channels, base_dir, pixel_cluster_col,
                                      num_pixel_clusters,
                                      pixel_data_dir='pixel_mat_data',
                                      num_fovs_subset=100, seed=42, keep_count=False):
    """Average channel values per pixel SOM / meta cluster
    (reference: pixel_cluster_utils.py:294-416).

    The per-FOV ``groupby(cluster)[channels].sum()`` / ``.size()`` and the sum over FOVs are one
    accumulating device reduction (binary64 sums, int64 counts); everything else -- validation,
    FOV sub-sampling with ``random.seed(seed)``, the error and warning texts, sorting and the
    ``count`` column -- follows the reference line by line.
    """
    verify_in_list(
        provided_cluster_col=[pixel_cluster_col],
        valid_cluster_cols=['pixel_som_cluster', 'pixel_meta_cluster']
    )

    if num_pixel_clusters is not None and num_pixel_clusters <= 0:
        raise ValueError("If set, number of pixel clusters desired must be a positive integer")

    if num_fovs_subset <= 0:
        raise ValueError("Number of fovs to subset must be a positive integer")

    if len(fovs) < num_fovs_subset:
        warnings.warn(
            'Provided num_fovs_subset=%d but only %d FOVs in dataset, '
            'subsetting just the %d FOVs' %
            (num_fovs_subset, len(fovs), len(fovs))
        )

    random.seed(seed)
    fovs_sub = random.sample(fovs, num_fovs_subset) if num_fovs_subset < len(fovs) else fovs

    channels = list(channels)
    sums = {}     # cluster id -> float64 [C]
    counts = {}   # cluster id -> int
    for fov in fovs_sub:
        try:
            fov_pixel_data = read_dataframe(
                os.path.join(base_dir, pixel_data_dir, fov + '.feather')
            )
        except (ArrowInvalid, OSError, IOError):
            print("The data for FOV %s has been corrupted, skipping" % fov)
            continue

        labels = fov_pixel_data[pixel_cluster_col].values
        if labels.size == 0:
            continue
        ids, inv = np.unique(labels, return_inverse=True)   # dense 1..len(ids) for the kernel
        fsum, fcnt = flowsom.cluster_sums(fov_pixel_data[channels].values,
                                          (inv + 1).astype(np.int32), len(ids))
        for pos, cid in enumerate(ids):
            if cid in sums:
                sums[cid] = sums[cid] + fsum[pos]
                counts[cid] += int(fcnt[pos])
            else:
                sums[cid] = fsum[pos].copy()
                counts[cid] = int(fcnt[pos])

    if not sums:
        # mirrors pd.concat([]) in the reference
        raise ValueError("No objects to concatenate")

    cluster_ids = sorted(sums)
    sum_count_totals = pd.DataFrame(np.stack([sums[cid] for cid in cluster_ids]), columns=channels)
    sum_count_totals.insert(0, pixel_cluster_col, cluster_ids)
    sum_count_totals['count'] = [counts[cid] for cid in cluster_ids]

    if num_pixel_clusters is not None and sum_count_totals.shape[0] < num_pixel_clusters:
        raise ValueError(
            'Averaged data contains just %d clusters out of %d. '
            'Average expression file not written. '
            'Consider increasing your num_fovs_subset value.' %
            (sum_count_totals.shape[0], num_pixel_clusters)
        )

    sum_count_totals[channels] = sum_count_totals[channels].div(sum_count_totals['count'], axis=0)
    sum_count_totals[pixel_cluster_col] = sum_count_totals[pixel_cluster_col].astype(int)
    sum_count_totals = sum_count_totals.sort_values(by=pixel_cluster_col)

    if not keep_count:
        sum_count_totals = sum_count_totals.drop('count', axis=1)

    return sum_count_totals


def find_fovs_missing_col(base_dir, data_dir, missing_col):
    """FOV names in ``data_dir`` without ``missing_col`` (reference: pixel_cluster_utils.py:419-478)."""
    data_path = os.path.join(base_dir, data_dir)
    temp_path = os.path.join(base_dir, data_dir + '_temp')

    validate_paths(data_path)

    if not os.path.exists(temp_path):
        fov_files = list_files(data_path, substrs='.feather')

        # read in a sample FOV, skipping potentially corrupted files
        i = 0
        fov_data = None
        while i < len(fov_files):
            try:
                fov_data = read_dataframe(os.path.join(data_path, fov_files[i]))
            except (ArrowInvalid, OSError, IOError):
                i += 1
                continue
            break

        if missing_col not in fov_data.columns.values:
            os.mkdir(temp_path)
            return remove_file_extensions(fov_files)
        else:
            return []
    else:
        data_files = set(list_files(data_path, substrs='.feather'))
        temp_files = set(list_files(temp_path, substrs='.feather'))
        leftover_files = list(data_files.difference(temp_files))
        return remove_file_extensions(leftover_files)
