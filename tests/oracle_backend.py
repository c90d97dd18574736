"""Oracle-backed stand-ins for the device entry points of ark_analysis_amd.flowsom (TEST INFRASTRUCTURE).

`install(patch)` swaps them in with the given setattr-like callable: pytest's ``monkeypatch.setattr`` in the
`som_backend` fixture (tests/conftest.py), plain ``setattr`` in the worker processes of the multi-rank CPU
tests.  This lets the host logic (file bookkeeping, restart rules, messages, rank sharding) run on a machine
without a GPU; the product never imports this module.
"""
import numpy as np
import torch


class OracleKernels:
    """Same interface as ark_analysis_amd.distributed.HipKernels, backed by oracle/pxsom_oracle.c."""

    def absmax(self, x):
        a = np.abs(x.numpy())
        a = a[np.isfinite(a)]
        return float(a.max()) if a.size else 0.0

    def begin(self, x, w, xdim, ydim, schedule, quantum=0.0):
        from ark_analysis_amd.schedule import resolve
        self.xdim, self.ydim, self.sch, self.quantum = xdim, ydim, resolve(schedule), float(quantum)
        k, c = w.shape
        self.w = [w.clone(), w.clone()]
        self.rings = [torch.zeros(k * (c + 1), dtype=torch.float64) for _ in range(3)]

    def ring(self, g):
        return self.rings[g % 3]

    def _update(self, w, g, total, alpha_range, radius_range):
        from tests import oracle_binding as ob
        from ark_analysis_amd.distributed import batch_schedule
        k, c = w.shape
        passes = total // self.sch.steps
        thr, alpha = batch_schedule(self.sch.position(g), passes * self.sch.phases, alpha_range, radius_range)
        st = self.rings[g % 3]
        return torch.from_numpy(ob.batch_update(w.numpy(), self.xdim, self.ydim, st[: k * c].view(k, c).numpy(),
                                                st[k * c:].numpy().astype(np.int64), thr, alpha))

    def steps(self, x, g0, g1, total, alpha_range, radius_range):
        from tests import oracle_binding as ob
        for g in range(g0, g1):
            if g > 0:
                self.w[g % 2] = self._update(self.w[(g - 1) % 2], g - 1, total, alpha_range, radius_range)
            w = self.w[g % 2]
            k, c = w.shape
            rows = self.sch.rows_of_step(x.shape[0], g)
            xn = np.ascontiguousarray(x.numpy()[rows], dtype=np.float64).reshape(-1, c)
            lab, _ = ob.map_data_to_nodes(w.numpy(), xn)
            s, cnt = ob.cluster_sums(ob.quantize(xn, self.quantum), lab, k)
            st = self.rings[g % 3]
            st[: k * c].copy_(torch.from_numpy(s.reshape(-1)))
            st[k * c:].copy_(torch.from_numpy(cnt.astype(np.float64)))

    def finish(self, steps_done, total, alpha_range, radius_range, w):
        g = steps_done - 1
        w.copy_(self._update(self.w[g % 2], g, total, alpha_range, radius_range))


def install(patch):
    """patch(module, "name", replacement) for every device entry point of ark_analysis_amd.flowsom."""
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

    patch(flowsom, "som", som)
    patch(flowsom, "map_data_to_nodes", map_data_to_nodes)
    patch(flowsom, "cluster_sums", cluster_sums)
    patch(flowsom, "pair_histogram", ob.pair_histogram)

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

    def fov_pixel_rows(img_hwc, sigma, thresh, nonzero_q=None, blocks=None):
        img_hwc = np.ascontiguousarray(img_hwc)
        f32 = img_hwc.dtype == np.float32
        h, w, c = img_hwc.shape
        blurred = ob.gaussian_blur_hwc(img_hwc, float(sigma), f32=f32)
        rows, kept = ob.rowsum_filter_normalize(blurred.reshape(h * w, c), float(thresh), sum_mode=2 if f32 else 0)
        rows = rows.astype(np.float32) if f32 else rows
        tail = () if blocks is None else (lambda: None,)        # "release" of a block that was never taken
        return ((rows, kept) if nonzero_q is None else (rows, kept, nonzero_quantiles(rows, nonzero_q))) + tail

    # generate_pixel_cluster_mask's relabel + scatter is plain numpy in the reference (utils/data_utils.py:532-553)
    def pixel_cluster_mask(row_index, column_index, labels, id_mapping, shape):
        img = np.zeros((int(shape[0]), int(shape[1])), dtype='int16')
        flat = img.ravel()
        ids = np.asarray([id_mapping[label] for label in np.asarray(labels).tolist()], dtype=np.int64)
        # the reference pins numpy < 1.24, where assigning a list of Python ints narrows silently
        flat[np.asarray(row_index) * img.shape[1] + np.asarray(column_index)] = ids.astype(np.int16)
        return flat.reshape(img.shape)

    patch(flowsom, "pixel_cluster_mask", pixel_cluster_mask)
    patch(flowsom, "fov_pixel_rows", fov_pixel_rows)
    patch(flowsom, "nonzero_quantiles", nonzero_quantiles)
    patch(flowsom, "positive_quantile_f32", positive_quantile_f32)
    patch(flowsom, "total_intensity_quantile_f32", total_intensity_quantile_f32)
    patch(flowsom, "_batch_backend", lambda: (torch.device("cpu"), OracleKernels()))
