"""``create_fov_pixel_data`` of ``ark.phenotyping.pixie_preprocessing``
(/root/reference/src/ark/phenotyping/pixie_preprocessing.py:18-80) on MI355X: per-channel Gaussian
blur, row-sum threshold, zero-row removal and row normalisation run as HIP kernels on the [H, W, C]
image in HBM; the DataFrame assembly and the seeded ``sample(frac=...)`` stay on the host exactly as
in the reference.  (TIFF loading and the cohort-level bookkeeping of ``create_pixel_matrix`` are out
of scope: SURVEY.md section 8 f.)"""
import numpy as np
import pandas as pd

from ..host_utils import natsort_key


def create_fov_pixel_data(fov, channels, img_data, seg_labels, pixel_thresh_val,
                          blur_factor=2, subset_proportion=0.1):
    """Preprocess pixel data for one fov; returns ``(pixel_mat, pixel_mat_subset)`` DataFrames with
    the reference's columns (channels, fov, row_index, column_index[, label])."""
    import torch
    from .. import _capi, som_device
    dev = _capi.require_gpu()
    channels.sort(key=natsort_key)                       # in place, like the reference (:44)
    h, w = img_data.shape[0], img_data.shape[1]
    # float32 images (what preprocess_fov passes for float32 TIFFs) keep float32 semantics end to end: scipy
    # stores each blur pass as float32, pandas sums and divides the float32 frame in binary32
    f32 = np.asarray(img_data).dtype == np.float32
    img = torch.from_numpy(np.ascontiguousarray(img_data[:, :, :len(channels)], dtype=np.float64)).to(dev)
    som_device.gaussian_blur_hwc(img, float(blur_factor), f32_semantics=f32)
    rows, kept = som_device.rowsum_filter_normalize(img.view(h * w, len(channels)), float(pixel_thresh_val),
                                                    f32_semantics=f32)
    kept_h = kept.cpu().numpy()
    values = rows.cpu().numpy()
    pixel_mat = pd.DataFrame(values.astype(np.float32) if f32 else values, columns=channels)
    pixel_mat['fov'] = fov
    pixel_mat['row_index'] = (kept_h // w).astype(np.int64)
    pixel_mat['column_index'] = (kept_h % w).astype(np.int64)
    if seg_labels is not None:
        pixel_mat['label'] = np.asarray(seg_labels).flatten()[kept_h]
    # subset the pixel matrix for training (global numpy RNG state, as the reference: :78)
    pixel_mat_subset = pixel_mat.sample(frac=subset_proportion)
    return pixel_mat, pixel_mat_subset
