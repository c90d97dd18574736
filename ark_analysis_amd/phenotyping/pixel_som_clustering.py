"""Pixel SOM pipeline functions -- drop-in for ``ark.phenotyping.pixel_som_clustering``
(/root/reference/src/ark/phenotyping/pixel_som_clustering.py).

Same signatures, validation, printed messages, restart logic and on-disk effects
(``<data_dir>_temp`` staging + directory swap) as the reference.  ``multiprocess=True`` is a
*hint* here: the GPU is the parallel resource, so FOVs are processed in-process, batch by
batch, with the reference's per-batch progress messages (the reference spawns a process pool
and pickles the SOM object into it, :257-271; a side effect of doing it in-process is that
``som_clusters_seen`` is no longer lost, a reference quirk noted in SURVEY.md section 3.2).
"""
import os
from shutil import move, rmtree
from typing import Any, Callable, Tuple

from pyarrow.lib import ArrowInvalid

from ..host_utils import (list_files, remove_file_extensions, validate_paths, verify_in_list,
                          verify_same_elements)
from . import cluster_helpers, pixel_cluster_utils
from .cluster_helpers import read_dataframe, write_dataframe


def train_pixel_som(fovs, channels, base_dir,
                    subset_dir='pixel_mat_subsetted',
                    norm_vals_name='post_rowsum_chan_norm.feather',
                    som_weights_name='pixel_som_weights.feather', xdim=10, ydim=10,
                    lr_start=0.05, lr_end=0.01, num_passes=1, seed=42,
                    overwrite=False):
    """Run the SOM training on the subsetted pixel data; saves the weights to
    ``base_dir/som_weights_name`` (reference: pixel_som_clustering.py:16-90)."""
    subsetted_path = os.path.join(base_dir, subset_dir)
    norm_vals_path = os.path.join(base_dir, norm_vals_name)
    som_weights_path = os.path.join(base_dir, som_weights_name)

    # NOTE: weights may or may not exist, that logic gets handled by PixelSOMCluster
    validate_paths([subsetted_path, norm_vals_path])

    # verify that all provided fovs exist in the folder
    files = list_files(subsetted_path, substrs='.feather')
    verify_in_list(provided_fovs=fovs,
                   subsetted_fovs=remove_file_extensions(files))

    # verify that all the provided channels exist in subsetted data
    sample_sub = read_dataframe(os.path.join(subsetted_path, files[0]))
    verify_in_list(provided_channels=channels,
                   subsetted_channels=sample_sub.columns.values)

    pixel_pysom = cluster_helpers.PixelSOMCluster(
        subsetted_path, norm_vals_path, som_weights_path, fovs, channels,
        num_passes=num_passes, xdim=xdim, ydim=ydim, lr_start=lr_start, lr_end=lr_end,
        seed=seed
    )

    print("Training SOM")
    pixel_pysom.train_som(overwrite=overwrite)

    return pixel_pysom


def run_pixel_som_assignment(pixel_data_path, pixel_pysom_obj, overwrite, num_parallel_pixels, fov):
    """Assign pixel SOM labels to one FOV file; returns ``(fov, status)`` with status 1 for a
    corrupted file (reference: pixel_som_clustering.py:93-136)."""
    fov_path = os.path.join(pixel_data_path, fov + '.feather')

    try:
        fov_data = read_dataframe(fov_path)
    # this indicates this fov file is corrupted
    except (ArrowInvalid, OSError, IOError):
        return fov, 1

    # if the overwrite flag was set in cluster_pixels, drop the pixel_som_cluster column
    if overwrite:
        fov_data = fov_data.drop(columns="pixel_som_cluster", errors="ignore")

    # assign the SOM labels to fov_data, overwrite flag indicates if data needs normalization
    fov_data = pixel_pysom_obj.assign_som_clusters(
        fov_data, normalize_data=not overwrite, num_parallel_pixels=num_parallel_pixels
    )

    # resave the data with the SOM cluster labels assigned
    temp_path = os.path.join(pixel_data_path + '_temp', fov + '.feather')
    write_dataframe(fov_data, temp_path, compression='uncompressed')

    return fov, 0


def cluster_pixels(fovs, base_dir, pixel_pysom, data_dir='pixel_mat_data',
                   multiprocess=False, batch_size=5, num_parallel_pixels=1000000,
                   overwrite=False):
    """Uses trained SOM weights to assign cluster labels on full pixel data; saves the data with
    labels to ``data_dir`` (reference: pixel_som_clustering.py:139-289)."""
    data_path = os.path.join(base_dir, data_dir)

    validate_paths([data_path])

    if pixel_pysom.weights is None:
        raise ValueError("Using untrained pixel_pysom object, please invoke train_pixel_som first")

    # verify that all provided fovs exist in the folder
    data_files = list_files(data_path, substrs='.feather')
    verify_in_list(provided_fovs=fovs,
                   subsetted_fovs=remove_file_extensions(data_files))

    # this will prevent reading in a corrupted sample_fov
    i = 0
    sample_fov = None
    while i < len(data_files):
        try:
            sample_fov = read_dataframe(os.path.join(base_dir, data_dir, data_files[i]))

            if "segmentation_label" in sample_fov.columns:
                sample_fov.rename(
                    columns={"segmentation_label": "label"},
                    inplace=True)
        except (ArrowInvalid, OSError, IOError):
            i += 1
            continue
        break

    # for verification purposes, drop the metadata columns
    cols_to_drop = ['fov', 'row_index', 'column_index']
    for col in ['label', 'pixel_som_cluster',
                'pixel_meta_cluster', 'pixel_meta_cluster_rename']:
        if col in sample_fov.columns.values:
            cols_to_drop.append(col)

    sample_fov = sample_fov.drop(
        columns=cols_to_drop
    )
    verify_same_elements(
        enforce_order=True,
        norm_vals_columns=pixel_pysom.norm_data.columns.values,
        pixel_data_columns=sample_fov.columns.values
    )

    # ensure the SOM weights columns are valid indexes
    verify_same_elements(
        enforce_order=True,
        pixel_som_weights_columns=pixel_pysom.weights.columns.values,
        pixel_data_columns=sample_fov.columns.values
    )

    # if overwrite flag set, run on all FOVs in data_dir, make sure to reset SOM clusters seen
    if overwrite:
        print('Overwrite flag set, reassigning SOM cluster labels to all FOVs')
        pixel_pysom.som_clusters_seen = set()
        os.mkdir(data_path + '_temp')
        fovs_list = remove_file_extensions(
            list_files(data_path, substrs='.feather')
        )
    # otherwise, only assign SOM clusters to FOVs that don't already have them
    else:
        fovs_list = pixel_cluster_utils.find_fovs_missing_col(
            base_dir, data_dir, 'pixel_som_cluster'
        )

    # make sure fovs_list only contain fovs that exist in the master fovs list specified
    fovs_list = list(set(fovs_list).intersection(fovs))

    # if there are no FOVs left without SOM labels don't run function
    if len(fovs_list) == 0:
        print("There are no more FOVs to assign SOM labels to, skipping")
        return

    # if SOM cluster labeling is only partially complete, inform the user of restart
    if len(fovs_list) < len(fovs):
        print("Restarting SOM label assignment from fov %s, "
              "%d fovs left to process" % (fovs_list[0], len(fovs_list)))

    fovs_processed = 0

    def fov_data_func(fov):
        return run_pixel_som_assignment(data_path, pixel_pysom, overwrite, num_parallel_pixels, fov)

    print("Mapping pixel data to SOM cluster labels")

    if multiprocess:
        # same batching and messages as the reference's Pool(batch_size) path, executed in-process
        for fov_batch in [fovs_list[i:(i + batch_size)]
                          for i in range(0, len(fovs_list), batch_size)]:
            fov_statuses = [fov_data_func(fov) for fov in fov_batch]

            for fs in fov_statuses:
                if fs[1] == 1:
                    print("The data for FOV %s has been corrupted, skipping" % fs[0])
                    fovs_processed -= 1

            fovs_processed += len(fov_batch)

            print("Processed %d fovs" % fovs_processed)
    else:
        for fov in fovs_list:
            fov_status = fov_data_func(fov)

            if fov_status[1] == 1:
                print("The data for FOV %s has been corrupted, skipping" % fov_status[0])
                fovs_processed -= 1

            fovs_processed += 1

            # update every 10 FOVs, or at the very end
            if fovs_processed % 10 == 0 or fovs_processed == len(fovs_list):
                print("Processed %d fovs" % fovs_processed)

    # remove the data directory and rename the temp directory to the data directory
    rmtree(data_path, onerror=_ignore_extended_attributes)
    move(data_path + '_temp', data_path)


def _ignore_extended_attributes(func: Callable, filename: str, exc_info: Tuple[Any, Any, Any]):
    """Ignore failures to remove extended attribute files (prefixed with "._")."""
    is_meta_file: bool = os.path.basename(filename).startswith("._")
    if not (func is os.unlink and is_meta_file):
        raise


def generate_som_avg_files(fovs, channels, base_dir, pixel_pysom, data_dir='pixel_data_dir',
                           pc_chan_avg_som_cluster_name='pixel_channel_avg_som_cluster.csv',
                           num_fovs_subset=100, require_all_som_clusters=True, seed=42,
                           overwrite=False):
    """Computes and saves the average channel expression across pixel SOM clusters
    (reference: pixel_som_clustering.py:308-371)."""
    som_cluster_avg_path = os.path.join(base_dir, pc_chan_avg_som_cluster_name)

    if pixel_pysom.weights is None:
        raise ValueError("Using untrained pixel_pysom object, please invoke train_som first")

    if os.path.exists(som_cluster_avg_path):
        if not overwrite:
            print("Already generated SOM cluster channel average file, skipping")
            return

        print("Overwrite flag set, regenerating SOM cluster channel average file")

    print("Computing average channel expression across pixel SOM clusters")
    pixel_channel_avg_som_cluster = pixel_cluster_utils.compute_pixel_cluster_channel_avg(
        fovs,
        channels,
        base_dir,
        'pixel_som_cluster',
        len(pixel_pysom.som_clusters_seen) if require_all_som_clusters else None,
        data_dir,
        num_fovs_subset=num_fovs_subset,
        seed=seed,
        keep_count=True
    )

    pixel_channel_avg_som_cluster.to_csv(
        som_cluster_avg_path,
        index=False
    )
