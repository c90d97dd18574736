"""SOM cluster objects -- the classes ``ark.phenotyping.cluster_helpers`` exposes for the Pixie SOM steps
(/root/reference/src/ark/phenotyping/cluster_helpers.py:52-416), re-implemented over the gfx950 kernels.

Public surface kept so notebooks and the reference's tests run unchanged: class names, constructor
arguments and defaults, the attributes callers read (``weights``, ``weights_path``, ``columns``,
``norm_data``, ``train_data``, ``cell_data``, ``fovs``, ``som_clusters_seen``), the warning texts and the
on-disk weights format (feather, one row per node, row k <-> label k + 1).  ``pyFlowSOM.som`` /
``pyFlowSOM.map_data_to_nodes`` are :func:`ark_analysis_amd.flowsom.som` /
:func:`~ark_analysis_amd.flowsom.map_data_to_nodes`.

Objects hold host state only (pandas / numpy) and therefore pickle; device buffers live for one call.

Deliberate differences:
* training tables are concatenated in natural-sorted FOV order -- the reference takes ``os.listdir``
  order (:211-215), which is file-system dependent, and the presentation order is an input of the SOM;
* rows are handed to the BMU search in positional blocks (the reference slices by index label,
  :154-156, which selects the same rows for the RangeIndex every table reader produces).
"""
import os
import pathlib
import warnings
from abc import ABC, abstractmethod
from typing import Iterator, List, Optional, Tuple

import numpy as np
import pandas as pd

from .. import flowsom
from ..fov_tables import FovTableDir, read_dataframe, write_dataframe  # noqa: F401  (re-exported)
from ..host_utils import validate_paths, verify_in_list


def _row_blocks(n_rows: int, block: int) -> Iterator[Tuple[int, int]]:
    """[start, stop) bounds of consecutive blocks of at most ``block`` rows."""
    start = 0
    while start < n_rows:
        yield start, min(start + block, n_rows)
        start += block


def _as_f64_matrix(frame: pd.DataFrame) -> np.ndarray:
    return np.ascontiguousarray(frame.to_numpy(dtype=np.float64))


class PixieSOMCluster(ABC):
    """What pixel and cell SOMs share: the grid, the schedule, the codebook file and the two kernel
    calls (reference base class: cluster_helpers.py:52-163)."""

    #: words of the "already trained" / "retraining" warnings; the subclasses fill them in
    _subject = ("SOM", "columns")

    @abstractmethod
    def __init__(self, weights_path: pathlib.Path, columns: List[str], num_passes: int = 1,
                 xdim: int = 10, ydim: int = 10, lr_start: float = 0.05, lr_end: float = 0.01,
                 seed=42):
        self.weights_path = weights_path
        self.columns = columns
        self.xdim, self.ydim = xdim, ydim
        self.lr_start, self.lr_end = lr_start, lr_end
        self.num_passes = num_passes
        self.seed = seed
        # a codebook written by an earlier session is picked up again
        self.weights: Optional[pd.DataFrame] = read_dataframe(weights_path) \
            if os.path.exists(weights_path) else None

    @abstractmethod
    def normalize_data(self) -> pd.DataFrame:
        """Scaling of the input columns; each subclass has its own rule."""

    # ---- training ---------------------------------------------------------------------------
    def _needs_training(self, overwrite: bool) -> bool:
        """The reference's three-way decision with its three warnings (:250-268, :374-393)."""
        kind, what = self._subject
        if overwrite:
            warnings.warn("Overwrite flag set, retraining SOM")
            return True
        if self.weights is None:
            return True
        if set(self.weights.columns.values) == set(self.columns):
            warnings.warn("%s SOM already trained on specified %s" % (kind, what))
            return False
        warnings.warn("New %s specified, retraining" % what)
        return True

    def train_som(self, data: pd.DataFrame):
        """Fit the xdim x ydim codebook on the rows of ``data`` (FlowSOM online rule, ``num_passes``
        passes, learning rate ``lr_start`` -> ``lr_end``) and store it next to ``weights_path``."""
        codebook = flowsom.som(data=_as_f64_matrix(data), xdim=self.xdim, ydim=self.ydim,
                               rlen=self.num_passes, alpha_range=(self.lr_start, self.lr_end),
                               seed=self.seed)
        nodes = self.xdim * self.ydim
        self.weights = pd.DataFrame(np.asarray(codebook).reshape(nodes, -1), columns=data.columns.values)
        write_dataframe(self.weights, self.weights_path, compression="uncompressed")

    # ---- assignment -------------------------------------------------------------------------
    def generate_som_clusters(self, external_data: pd.DataFrame,
                              num_parallel_obs: int = 1000000) -> np.ndarray:
        """1-based best-matching-unit label of every row of ``external_data`` (first minimum wins),
        searched in blocks of ``num_parallel_obs`` rows."""
        if num_parallel_obs <= 0:
            raise ValueError("num_parallel_obs specified needs to be greater than 0")

        trained_on = self.weights.columns.values
        verify_in_list(weights_cols=trained_on, external_data_cols=external_data.columns.values)

        codebook = _as_f64_matrix(self.weights)
        features = external_data[list(trained_on)]  # also puts the columns in codebook order
        found = [flowsom.map_data_to_nodes(codebook, _as_f64_matrix(features.iloc[lo:hi]))[0]
                 for lo, hi in _row_blocks(len(external_data), num_parallel_obs)]
        # an image without retained pixels gives an empty (float) vector, like the reference
        return np.concatenate(found) if found else np.empty(0)


class PixelSOMCluster(PixieSOMCluster):
    """Pixel SOM: trains on the sub-sampled tables of ``pixel_subset_folder``, values divided by the
    per-channel 99.9 % row of ``norm_vals_path`` (reference: cluster_helpers.py:166-301)."""

    _subject = ("Pixel", "markers")

    def __init__(self, pixel_subset_folder: pathlib.Path, norm_vals_path: pathlib.Path,
                 weights_path: pathlib.Path, fovs: List[str], columns: List[str],
                 num_passes: int = 1, xdim: int = 10, ydim: int = 10,
                 lr_start: float = 0.05, lr_end: float = 0.01, seed=42):
        super().__init__(weights_path, columns, num_passes, xdim, ydim, lr_start, lr_end, seed)
        validate_paths([norm_vals_path, pixel_subset_folder])

        self.fovs = fovs
        self.norm_data = read_dataframe(norm_vals_path)
        self.som_clusters_seen = set()

        wanted = set(fovs)
        subset = FovTableDir(pixel_subset_folder)
        parts = [subset.load(fov) for fov in subset.fovs() if fov in wanted]
        # the training matrix is only ever used normalised
        self.train_data = self.normalize_data(pd.concat(parts))

    def normalize_data(self, external_data: pd.DataFrame) -> pd.DataFrame:
        """Copy of ``external_data`` with every channel of ``norm_data`` divided by its norm value."""
        channels = self.norm_data.columns.values
        verify_in_list(norm_data_cols=channels, external_data_cols=external_data.columns.values)

        scaled = external_data.copy()
        divisors = self.norm_data.iloc[0].to_numpy(dtype=np.float64)
        scaled[channels] = scaled[channels].to_numpy() / divisors   # IEEE division per element
        return scaled

    def train_som(self, overwrite=False):
        if self._needs_training(overwrite):
            super().train_som(self.train_data[self.columns])

    def assign_som_clusters(self, external_data: pd.DataFrame, normalize_data: bool = True,
                            num_parallel_pixels: int = 1000000) -> pd.DataFrame:
        """``external_data`` (normalised unless told otherwise) plus a ``pixel_som_cluster`` column;
        the labels met are remembered in ``som_clusters_seen``."""
        table = self.normalize_data(external_data) if normalize_data else external_data.copy()
        labels = self.generate_som_clusters(table, num_parallel_obs=num_parallel_pixels)
        table["pixel_som_cluster"] = labels
        self.som_clusters_seen.update(np.unique(labels).tolist())
        return table


class CellSOMCluster(PixieSOMCluster):
    """Cell SOM over a cell x feature table held in memory; features are scaled by their own 99.9 %
    quantile of non-zero values (reference: cluster_helpers.py:304-416)."""

    _subject = ("Cell", "columns")

    def __init__(self, cell_data: pd.DataFrame, weights_path: pathlib.Path,
                 fovs: List[str], columns: List[str], num_passes: int = 1,
                 xdim: int = 10, ydim: int = 10, lr_start: float = 0.05, lr_end: float = 0.01,
                 seed=42, normalize=True):
        super().__init__(weights_path, columns, num_passes, xdim, ydim, lr_start, lr_end, seed)
        self.fovs = fovs
        in_cohort = cell_data["fov"].isin(fovs)
        self.cell_data = cell_data[in_cohort].reset_index(drop=True)
        if normalize:
            self.normalize_data()

    def normalize_data(self):
        """In place: each training column divided by the 0.999 quantile of its non-zero entries."""
        block = self.cell_data[self.columns]
        caps = block.where(block != 0).quantile(q=0.999, axis=0)   # zeros -> NaN -> ignored
        self.cell_data[self.columns] = block.div(caps)

    def train_som(self, overwrite=False):
        if self._needs_training(overwrite):
            super().train_som(self.cell_data[self.columns])

    def assign_som_clusters(self, num_parallel_cells=1000000) -> pd.DataFrame:
        """Adds ``cell_som_cluster`` to ``cell_data`` and returns the table."""
        self.cell_data["cell_som_cluster"] = self.generate_som_clusters(
            self.cell_data[self.columns], num_parallel_obs=num_parallel_cells)
        return self.cell_data
