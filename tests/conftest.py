import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from tests import oracle_binding
    oracle_binding.build()
    return oracle_binding


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from ark_analysis_amd import _capi
    _capi.lib()
    return torch.device("cuda:0")


@pytest.fixture(params=[pytest.param("oracle"), pytest.param("hip", marks=pytest.mark.gpu)])
def som_backend(request, monkeypatch):
    """Runs a host-logic test twice: on CPU with the device entry points of
    ark_analysis_amd.flowsom swapped for the oracle (test infrastructure; `-m "not gpu"`), and on
    the GPU box with the real HIP path (`-m gpu`)."""
    if request.param == "hip":
        import torch
        if not torch.cuda.is_available():
            pytest.skip("no HIP device")
        return "hip"
    import numpy as np
    from tests import oracle_binding as ob
    from ark_analysis_amd import flowsom

    def som(data, xdim=10, ydim=10, rlen=10, alpha_range=(0.05, 0.01), radius_range=None, distf=2,
            nodes=None, importance=None, seed=None):
        data = np.ascontiguousarray(data, dtype=np.float64)
        init_idx, order = flowsom.som_init_and_order(data.shape[0], xdim * ydim, rlen, seed)
        codes = data[init_idx].copy() if nodes is None else np.array(nodes, dtype=np.float64)
        if radius_range is None:
            radius_range = flowsom.default_radius_range(xdim, ydim)
        return ob.som_online(data, codes, xdim, ydim, rlen, alpha_range, radius_range, order)

    def map_data_to_nodes(nodes, newdata, distf=2):
        return ob.map_data_to_nodes(np.asarray(nodes, dtype=np.float64),
                                    np.asarray(newdata, dtype=np.float64))

    def cluster_sums(data, labels, k):
        return ob.cluster_sums(np.asarray(data, dtype=np.float64), labels, int(k))

    monkeypatch.setattr(flowsom, "som", som)
    monkeypatch.setattr(flowsom, "map_data_to_nodes", map_data_to_nodes)
    monkeypatch.setattr(flowsom, "cluster_sums", cluster_sums)
    monkeypatch.setattr(flowsom, "pair_histogram", ob.pair_histogram)

    # the TIFF-side percentiles ARE numpy calls in the reference (pixel_cluster_utils.py:41-51, :96-103)
    def positive_quantile_f32(image, q):
        image = np.asarray(image)

        def one(plane):
            kept = plane[plane > 0]
            return np.quantile(kept, q) if kept.size else np.float32("nan")
        if image.ndim == 2:
            return one(image)
        return np.array([one(image[:, :, j]) for j in range(image.shape[2])])

    def total_intensity_quantile_f32(image_hwc, norm, q):
        return np.quantile(np.sum(image_hwc / np.asarray(norm).reshape([1, 1, -1]), axis=-1), q)

    def nonzero_quantiles(matrix, q):
        m = np.asarray(matrix, dtype=np.float64)
        return np.array([ob.quantile_nonzero(np.ascontiguousarray(m[:, j]), q, 0) for j in range(m.shape[1])])

    def fov_pixel_rows(img_hwc, sigma, thresh, nonzero_q=None):
        img_hwc = np.ascontiguousarray(img_hwc)
        f32 = img_hwc.dtype == np.float32
        h, w, c = img_hwc.shape
        blurred = ob.gaussian_blur_hwc(img_hwc, float(sigma), f32=f32)
        rows, kept = ob.rowsum_filter_normalize(blurred.reshape(h * w, c), float(thresh), sum_mode=2 if f32 else 0)
        rows = rows.astype(np.float32) if f32 else rows
        return (rows, kept) if nonzero_q is None else (rows, kept, nonzero_quantiles(rows, nonzero_q))

    # generate_pixel_cluster_mask's relabel + scatter is plain numpy in the reference (utils/data_utils.py:532-553)
    def pixel_cluster_mask(row_index, column_index, labels, id_mapping, shape):
        img = np.zeros((int(shape[0]), int(shape[1])), dtype='int16')
        flat = img.ravel()
        ids = np.asarray([id_mapping[label] for label in np.asarray(labels).tolist()], dtype=np.int64)
        # the reference pins numpy < 1.24, where assigning a list of Python ints narrows silently
        flat[np.asarray(row_index) * img.shape[1] + np.asarray(column_index)] = ids.astype(np.int16)
        return flat.reshape(img.shape)

    monkeypatch.setattr(flowsom, "pixel_cluster_mask", pixel_cluster_mask)
    monkeypatch.setattr(flowsom, "fov_pixel_rows", fov_pixel_rows)
    monkeypatch.setattr(flowsom, "nonzero_quantiles", nonzero_quantiles)
    monkeypatch.setattr(flowsom, "positive_quantile_f32", positive_quantile_f32)
    monkeypatch.setattr(flowsom, "total_intensity_quantile_f32", total_intensity_quantile_f32)
    return "oracle"
