"""``create_fov_pixel_data`` of ``ark.phenotyping.pixie_preprocessing``
(/root/reference/src/ark/phenotyping/pixie_preprocessing.py:18-80) on MI355X: per-channel Gaussian
blur, row-sum threshold, zero-row removal and row normalisation run as HIP kernels on the [H, W, C]
image in HBM; the DataFrame assembly and the seeded ``sample(frac=...)`` stay on the host exactly as
in the reference.  ``preprocess_fov`` / ``create_pixel_matrix`` (:83-456) wrap it for a cohort of TIFF
folders (Pillow instead of scikit-image / alpineer, which this image lacks); float32 TIFFs keep the
reference's float32 arithmetic end to end."""
import os
import shutil

import numpy as np
import pandas as pd

from .. import flowsom, image_io
from ..fov_tables import FovTableDir, read_dataframe, write_dataframe
from ..host_utils import natsort_key, validate_paths, verify_in_list
from . import pixel_cluster_utils


def create_fov_pixel_data(fov, channels, img_data, seg_labels, pixel_thresh_val,
                          blur_factor=2, subset_proportion=0.1):
    """Preprocess pixel data for one fov; returns ``(pixel_mat, pixel_mat_subset)`` DataFrames with
    the reference's columns (channels, fov, row_index, column_index[, label])."""
    channels.sort(key=natsort_key)                       # in place, like the reference (:44)
    w = img_data.shape[1]
    # float32 images (what preprocess_fov passes for float32 TIFFs) keep float32 semantics end to end: scipy
    # stores each blur pass as float32, pandas sums and divides the float32 frame in binary32
    values, kept_h = flowsom.fov_pixel_rows(np.asarray(img_data)[:, :, :len(channels)], blur_factor, pixel_thresh_val)
    pixel_mat = pd.DataFrame(values, columns=channels)
    pixel_mat['fov'] = fov
    pixel_mat['row_index'] = (kept_h // w).astype(np.int64)
    pixel_mat['column_index'] = (kept_h % w).astype(np.int64)
    if seg_labels is not None:
        pixel_mat['label'] = np.asarray(seg_labels).flatten()[kept_h]
    # subset the pixel matrix for training (global numpy RNG state, as the reference: :78)
    pixel_mat_subset = pixel_mat.sample(frac=subset_proportion)
    return pixel_mat, pixel_mat_subset


# ---- cohort level: TIFF folders -> per-FOV pixel tables + normalisation files ----------------------------
# (reference: preprocess_fov, pixie_preprocessing.py:83-198, and create_pixel_matrix, :201-456.)

_PER_FOV_QUANTILES = "channel_norm_post_rownorm_perfov.csv"


def _read_segmentation(seg_dir, fov, seg_suffix):
    from PIL import Image
    with Image.open(os.path.join(seg_dir, fov + seg_suffix)) as im:
        return np.array(im)


def preprocess_fov(base_dir, tiff_dir, data_dir, subset_dir, seg_dir, seg_suffix,
                   img_sub_folder, is_mibitiff, channels, blur_factor,
                   subset_proportion, pixel_thresh_val, seed, channel_norm_df, fov):
    """One FOV from TIFFs to tables: load the channel images, divide by the pre-row-norm channel values,
    run :func:`create_fov_pixel_data` (seeded), write ``<data_dir>/<fov>.feather`` and
    ``<subset_dir>/<fov>.feather``; returns the full table (the caller needs its 99.9 % values)."""
    if is_mibitiff:
        raise NotImplementedError("multi-page MIBItiff input is not built; export single-channel TIFFs")
    verify_in_list(provided_chans=channels,
                   pixel_mat_chans=image_io.channel_names(tiff_dir, fov, img_sub_folder))
    labels = _read_segmentation(seg_dir, fov, seg_suffix) if seg_dir is not None else None

    stack = image_io.read_channels(tiff_dir, fov, channels, img_sub_folder).astype(np.float32)
    stack = stack / np.array(channel_norm_df.iloc[0].values).reshape([1, 1, -1])   # float32 / float32 stays float32

    np.random.seed(seed)
    full, subset = create_fov_pixel_data(fov=fov, channels=channels, img_data=stack, seg_labels=labels,
                                         pixel_thresh_val=pixel_thresh_val, blur_factor=blur_factor,
                                         subset_proportion=subset_proportion)
    write_dataframe(full, os.path.join(base_dir, data_dir, fov + ".feather"), compression='uncompressed')
    write_dataframe(subset, os.path.join(base_dir, subset_dir, fov + ".feather"), compression='uncompressed')
    return full


def _nonzero_quantile_row(table, feature_cols, q, fov):
    """``table[feature_cols].replace(0, nan).quantile(q)`` as a Series named ``fov`` (index ``channel``).
    Binary64 with pandas' effective q, also for float32 tables: on the frames ``create_fov_pixel_data``
    builds (one block per channel after ``normalize_rows``) pandas returns the binary64 percentile uncast."""
    got = flowsom.nonzero_quantiles(table[feature_cols].to_numpy(dtype=np.float64), (q * 100) / 100)
    return pd.Series(got, index=pd.Index(feature_cols, name="channel"), name=fov)


def create_pixel_matrix(fovs, channels, base_dir, tiff_dir, seg_dir,
                        img_sub_folder="TIFs", seg_suffix='_whole_cell.tiff',
                        pixel_output_dir='pixel_output_dir',
                        data_dir='pixel_mat_data',
                        subset_dir='pixel_mat_subsetted',
                        norm_vals_name_pre_rownorm='channel_norm_pre_rownorm.feather',
                        norm_vals_name_post_rownorm='channel_norm_post_rownorm.feather',
                        pixel_thresh_name='pixel_thresh.feather',
                        channel_percentile_pre_rownorm=0.99, channel_percentile_post_rownorm=0.999,
                        is_mibitiff=False, blur_factor=2, subset_proportion=0.1, seed=42,
                        multiprocess=False, batch_size=5):
    """Blur, threshold and row-normalise every FOV of the cohort, write the full and the sub-sampled pixel
    tables, and derive the three normalisation files (pre-row-norm channel values, pixel threshold,
    post-row-norm 99.9 % values).  Restartable: FOVs whose tables (and per-FOV 99.9 % values) already
    exist are skipped; a changed channel list resets the cohort.  ``multiprocess`` only changes the
    progress lines (FOVs are processed by the GPU of the calling process either way)."""
    channels.sort(key=natsort_key)
    if subset_proportion <= 0 or subset_proportion > 1:
        raise ValueError('Invalid subset percentage entered: must be in (0, 1]')
    out_root = os.path.join(base_dir, pixel_output_dir)
    validate_paths([base_dir, tiff_dir, out_root])

    data_root, subset_root = os.path.join(base_dir, data_dir), os.path.join(base_dir, subset_dir)
    for folder in (data_root, subset_root):
        os.makedirs(folder, exist_ok=True)
    pre_norm_file = os.path.join(out_root, norm_vals_name_pre_rownorm)
    thresh_file = os.path.join(out_root, pixel_thresh_name)
    per_fov_file = os.path.join(data_root, _PER_FOV_QUANTILES)

    # a different channel selection invalidates everything derived so far
    if os.path.exists(pre_norm_file) and set(read_dataframe(pre_norm_file).columns.values) != set(channels):
        print("New channels provided: overwriting whole cohort")
        for folder in (data_root, subset_root):
            shutil.rmtree(folder)
            os.mkdir(folder)
        os.remove(pre_norm_file)
        os.remove(thresh_file)

    # finished = both tables on disk; their per-FOV 99.9 % values must be on record as well
    finished = set(FovTableDir(data_root).fovs()) & set(FovTableDir(subset_root).fovs())
    todo = set(fovs) - finished
    if not todo:
        print("There are no more FOVs to preprocess, skipping")
        return
    per_fov = pd.read_csv(per_fov_file, index_col="channel") if os.path.exists(per_fov_file) else pd.DataFrame()
    todo = list(todo | (set(fovs) - set(per_fov.columns)))
    if len(todo) < len(fovs):
        print("Restarting preprocessing from FOV %s, "
              "%d fovs left to process" % (todo[0], len(todo)))

    pixel_cluster_utils.check_for_modified_channels(tiff_dir=tiff_dir, test_fov=fovs[0],
                                                    img_sub_folder=img_sub_folder, channels=channels)

    if os.path.exists(pre_norm_file):
        pre_norm = read_dataframe(pre_norm_file)
    else:
        pre_norm = pixel_cluster_utils.calculate_channel_percentiles(
            tiff_dir=tiff_dir, fovs=fovs, channels=channels, img_sub_folder=img_sub_folder,
            percentile=channel_percentile_pre_rownorm)
        write_dataframe(pre_norm, pre_norm_file, compression='uncompressed')

    if os.path.exists(thresh_file):
        pixel_thresh_val = read_dataframe(thresh_file)['pixel_thresh_val'].values[0]
    else:
        pixel_thresh_val = pixel_cluster_utils.calculate_pixel_intensity_percentile(
            tiff_dir=tiff_dir, fovs=fovs, channels=channels, img_sub_folder=img_sub_folder,
            channel_percentiles=pre_norm)
        write_dataframe(pd.DataFrame({'pixel_thresh_val': [pixel_thresh_val]}), thresh_file,
                        compression='uncompressed')

    not_features = ['fov', 'row_index', 'column_index'] + (['label'] if seg_dir else [])
    group = batch_size if multiprocess else 1
    done = 0
    for start in range(0, len(todo), group):
        names = todo[start:start + group]
        for fov in names:
            table = preprocess_fov(base_dir, tiff_dir, data_dir, subset_dir, seg_dir, seg_suffix, img_sub_folder,
                                   is_mibitiff, channels, blur_factor, subset_proportion, pixel_thresh_val, seed,
                                   pre_norm, fov)
            features = [c for c in table.columns if c not in not_features]
            row = _nonzero_quantile_row(table, features, channel_percentile_post_rownorm, fov)
            per_fov = per_fov.merge(row, how="outer", left_index=True, right_index=True)
            per_fov.to_csv(per_fov_file)   # after every FOV: an interrupted run keeps what it has
        done += len(names)
        if multiprocess or done % 10 == 0 or done == len(todo):
            print("Processed %d fovs" % done)

    # cohort value per channel = mean of the per-FOV values; channels in natural order
    cohort = pd.DataFrame(per_fov.mean(axis=1))
    cohort = cohort.loc[sorted(cohort.index, key=natsort_key)]
    write_dataframe(cohort.T, os.path.join(base_dir, norm_vals_name_post_rownorm), compression='uncompressed')
    os.remove(per_fov_file)
