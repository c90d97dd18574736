"""The functions of ``ark.utils.data_utils`` that sit directly behind the pixel labels
(/root/reference/src/ark/utils/data_utils.py:476-555; SURVEY.md section 8 f, rank 4): a FOV's pixel table with
SOM / meta cluster labels -> an ``[H, W]`` int16 image of cluster ids.  The relabel (label -> cluster_id) and
the scatter run on the device; path checks, the mapping table and the table read stay on the host.  Around it, the
cohort loop that saves one mask per FOV (``generate_and_save_pixel_cluster_masks``, :558-635) and ``save_fov_mask``
(:32-68)."""
import os

import numpy as np
import pandas as pd

from .. import distributed, flowsom, image_io
from ..fov_tables import read_table
from ..host_utils import validate_paths, verify_in_list


def generate_pixel_cluster_mask(fov, base_dir, tiff_dir, chan_file_path,
                                pixel_data_dir, cluster_mapping,
                                pixel_cluster_col='pixel_meta_cluster'):
    """For a fov, create a mask labeling each pixel with its SOM or meta cluster id.

    ``chan_file_path`` (relative to ``tiff_dir``) names a sample channel image that fixes the mask's size;
    ``cluster_mapping`` is the DataFrame that maps ``pixel_cluster_col`` values to ``cluster_id``.  Pixels the
    table does not list stay 0."""
    table_dir = os.path.join(base_dir, pixel_data_dir)
    validate_paths([tiff_dir, os.path.join(tiff_dir, chan_file_path), table_dir])
    verify_in_list(provided_cluster_col=[pixel_cluster_col],
                   valid_cluster_cols=['pixel_som_cluster', 'pixel_meta_cluster'])
    verify_in_list(provided_fov_file=[fov + '.feather'], consensus_fov_files=os.listdir(table_dir))

    sample = np.squeeze(image_io.read_image(os.path.join(tiff_dir, chan_file_path)))
    table = read_table(os.path.join(table_dir, fov + '.feather'))

    def column(name):
        return table.column(name).to_numpy()
    labels = column(pixel_cluster_col).astype(int)        # "ensure integer display and not float"

    # later rows of the mapping win for a repeated key, as in dict(zip(...))
    pairs = cluster_mapping.drop_duplicates()[[pixel_cluster_col, 'cluster_id']]
    id_mapping = dict(zip(pairs[pixel_cluster_col], pairs['cluster_id']))
    return flowsom.pixel_cluster_mask(column('row_index'), column('column_index'), labels, id_mapping,
                                      (sample.shape[0], sample.shape[1]))


def save_fov_mask(fov, data_dir, mask_data, sub_dir=None, name_suffix=''):
    """Saves a cluster mask as ``<data_dir>/[<sub_dir>/]<fov><name_suffix>.tiff``."""
    validate_paths(data_dir)
    folder = os.path.join(data_dir, sub_dir or '')
    os.makedirs(folder, exist_ok=True)
    mask = np.asarray(mask_data)
    path = os.path.join(folder, fov + name_suffix + '.tiff')
    if mask.dtype.kind in "biu" and mask.dtype not in (np.uint8, np.uint16, np.int16, np.int32):
        # dtypes the baseline TIFF writer has no sample format for (bool, uint32, int64 ... masks: the reference's tifffile
        # takes them all): the narrowest one it has that holds every value -- pixel values are unchanged
        lo, hi = (int(mask.min()), int(mask.max())) if mask.size else (0, 0)
        for dt in (np.uint8, np.uint16, np.int16, np.int32):
            if np.iinfo(dt).min <= lo and hi <= np.iinfo(dt).max:
                mask = mask.astype(dt)
                break
        else:
            raise ValueError("mask values [%d, %d] do not fit a 32-bit TIFF sample" % (lo, hi))
    elif mask.dtype == np.float64:
        as32 = mask.astype(np.float32)
        if not np.array_equal(as32.astype(np.float64), mask, equal_nan=True):
            raise ValueError("float64 mask with values float32 cannot hold: save it with a full TIFF library")
        mask = as32
    image_io.write_image(path, mask)


def generate_and_save_pixel_cluster_masks(fovs, base_dir, save_dir, tiff_dir, chan_file, pixel_data_dir,
                                          cluster_id_to_name_path, pixel_cluster_col='pixel_meta_cluster',
                                          sub_dir=None, name_suffix=''):
    """One cluster-id mask per FOV, saved under ``save_dir``.  The cluster -> name table at
    ``cluster_id_to_name_path`` (the remapping GUI's output) gets a ``cluster_id`` column -- 1, 2, ... over the
    distinct ``pixel_cluster_col`` values in ascending order -- and is rewritten in place; those ids are what the
    masks hold.  ``chan_file``: a channel image inside every FOV folder, for the mask's size."""
    # under a process group (torchrun) rank 0 rewrites the table, the FOVs are dealt out by rank
    rank, world = distributed.init_from_env()
    mapping = None
    if rank == 0:
        names = pd.read_csv(cluster_id_to_name_path)
        ids = names[[pixel_cluster_col]].drop_duplicates().sort_values(by=[pixel_cluster_col])
        ids["cluster_id"] = list(range(1, len(ids) + 1))
        mapping = names.drop(columns="cluster_id", errors="ignore").merge(ids, on=[pixel_cluster_col], how="left")
        mapping.to_csv(cluster_id_to_name_path, index=False)
    mapping = distributed.broadcast_object(mapping, 0)
    for fov in distributed.shard(fovs, rank, world):
        mask = generate_pixel_cluster_mask(fov=fov, base_dir=base_dir, tiff_dir=tiff_dir,
                                           chan_file_path=os.path.join(fov, chan_file), pixel_data_dir=pixel_data_dir,
                                           cluster_mapping=mapping, pixel_cluster_col=pixel_cluster_col)
        save_fov_mask(fov, data_dir=save_dir, mask_data=mask, sub_dir=sub_dir, name_suffix=name_suffix)
    distributed.barrier()
