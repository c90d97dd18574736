"""SOM cluster objects -- drop-in for ``ark.phenotyping.cluster_helpers``
(/root/reference/src/ark/phenotyping/cluster_helpers.py:52-416).

Same classes, constructor signatures, attributes, warning strings and on-disk formats as the
reference; ``pyFlowSOM.som`` / ``pyFlowSOM.map_data_to_nodes`` are replaced by the gfx950
implementations in :mod:`ark_analysis_amd.flowsom`.  Objects hold only host state (pandas /
numpy), so they pickle like the reference's; device buffers are created per call.

Differences, all deliberate and documented in DESIGN.md:
* feather I/O goes through ``pyarrow.feather`` (the ``feather-format`` package is that re-export).
* training tables are concatenated in natural-sorted FOV order (the reference takes
  ``os.listdir`` order, cluster_helpers.py:211-215).
* assignment batches are sliced positionally (the reference's label slice ``.loc[i:i+B-1]``,
  :154-156, selects the same rows for the RangeIndex every reader produces).
"""
import os
import pathlib
import warnings
from abc import ABC, abstractmethod
from typing import List

import numpy as np
import pandas as pd
import pyarrow as pa

from .. import flowsom
from ..host_utils import list_files, validate_paths, verify_in_list


def read_dataframe(path) -> pd.DataFrame:
    """feather.read_dataframe: a Feather V2 file is an Arrow IPC file.  A corrupted file raises
    pyarrow.lib.ArrowInvalid / OSError, which the pipeline functions catch like the reference."""
    with pa.OSFile(str(path), "rb") as f:
        return pa.ipc.open_file(f).read_all().to_pandas()


def write_dataframe(df: pd.DataFrame, path, compression="uncompressed") -> None:
    """feather.write_dataframe(df, path, compression='uncompressed')."""
    table = pa.Table.from_pandas(df, preserve_index=None)
    codec = None if compression in (None, "uncompressed") else compression
    opts = pa.ipc.IpcWriteOptions(compression=codec)
    with pa.OSFile(str(path), "wb") as f:
        with pa.ipc.new_file(f, table.schema, options=opts) as w:
            w.write_table(table, max_chunksize=65536)


class PixieSOMCluster(ABC):
    @abstractmethod
    def __init__(self, weights_path: pathlib.Path, columns: List[str], num_passes: int = 1,
                 xdim: int = 10, ydim: int = 10, lr_start: float = 0.05, lr_end: float = 0.01,
                 seed=42):
        """Generic SOM runner (reference: cluster_helpers.py:52-87)."""
        self.weights_path = weights_path
        self.weights = None if not os.path.exists(weights_path) else read_dataframe(weights_path)
        self.columns = columns
        self.num_passes = num_passes
        self.xdim = xdim
        self.ydim = ydim
        self.lr_start = lr_start
        self.lr_end = lr_end
        self.seed = seed

    @abstractmethod
    def normalize_data(self) -> pd.DataFrame:
        """Normalisation of the input data (implemented by the subclasses)."""

    def train_som(self, data: pd.DataFrame):
        """Trains the SOM on ``data`` and saves the weights (reference: cluster_helpers.py:98-116)."""
        som_weights = flowsom.som(
            data=data.values.astype(np.float64), xdim=self.xdim, ydim=self.ydim,
            rlen=self.num_passes, alpha_range=(self.lr_start, self.lr_end), seed=self.seed
        )
        # ensure dimensions of weights are flattened
        som_weights = np.reshape(som_weights, (self.xdim * self.ydim, som_weights.shape[-1]))
        self.weights = pd.DataFrame(som_weights, columns=data.columns.values)
        write_dataframe(self.weights, self.weights_path, compression='uncompressed')

    def generate_som_clusters(self, external_data: pd.DataFrame,
                              num_parallel_obs: int = 1000000) -> np.ndarray:
        """BMU label (1-based) of every row (reference: cluster_helpers.py:118-163)."""
        if num_parallel_obs <= 0:
            raise ValueError("num_parallel_obs specified needs to be greater than 0")

        # subset on just the weights columns prior to SOM cluster mapping
        weights_cols = self.weights.columns.values
        verify_in_list(
            weights_cols=weights_cols,
            external_data_cols=external_data.columns.values
        )

        cluster_labels = []
        weights = self.weights.values.astype(np.float64)
        # NOTE: indexing by weights_cols also orders the columns like self.weights
        data = external_data[list(weights_cols)]
        for i in np.arange(0, external_data.shape[0], num_parallel_obs):
            cluster_labels.append(flowsom.map_data_to_nodes(
                weights, data.iloc[i:i + num_parallel_obs].values.astype(np.float64)
            )[0])

        # if no pixels in the image, return empty array
        if not cluster_labels:
            return np.empty(0)
        return np.concatenate(cluster_labels)


class PixelSOMCluster(PixieSOMCluster):
    def __init__(self, pixel_subset_folder: pathlib.Path, norm_vals_path: pathlib.Path,
                 weights_path: pathlib.Path, fovs: List[str], columns: List[str],
                 num_passes: int = 1, xdim: int = 10, ydim: int = 10,
                 lr_start: float = 0.05, lr_end: float = 0.01, seed=42):
        """Pixel SOM cluster object (reference: cluster_helpers.py:166-221)."""
        super().__init__(
            weights_path, columns, num_passes, xdim, ydim, lr_start, lr_end, seed
        )

        # path validation
        validate_paths([norm_vals_path, pixel_subset_folder])

        # load the normalization values in
        self.norm_data = read_dataframe(norm_vals_path)

        # define the fovs used
        self.fovs = fovs

        # list all the files in pixel_subset_folder and load them to train_data
        fov_files = list_files(pixel_subset_folder, substrs='.feather')
        self.train_data = pd.concat(
            [read_dataframe(os.path.join(pixel_subset_folder, fov)) for fov in fov_files
             if os.path.splitext(fov)[0] in fovs]
        )

        # we can just normalize train_data now since that's what we'll be training on
        self.train_data = self.normalize_data(self.train_data)

        # define each SOM cluster seen
        self.som_clusters_seen = set()

    def normalize_data(self, external_data: pd.DataFrame) -> pd.DataFrame:
        """``external_data[norm cols] / norm_data`` (reference: cluster_helpers.py:223-248)."""
        verify_in_list(
            norm_data_cols=self.norm_data.columns.values,
            external_data_cols=external_data.columns.values
        )

        norm_data_cols = self.norm_data.columns.values
        external_data_norm = external_data.copy()
        external_data_norm[norm_data_cols] = external_data_norm[norm_data_cols].div(
            self.norm_data.iloc[0], axis=1
        )
        return external_data_norm

    def train_som(self, overwrite=False):
        """Trains the SOM using ``train_data`` (reference: cluster_helpers.py:250-268)."""
        if overwrite:
            warnings.warn('Overwrite flag set, retraining SOM')
        elif self.weights is not None:
            if set(self.weights.columns.values) == set(self.columns):
                warnings.warn('Pixel SOM already trained on specified markers')
                return
            warnings.warn('New markers specified, retraining')

        super().train_som(self.train_data[self.columns])

    def assign_som_clusters(self, external_data: pd.DataFrame,
                            normalize_data: bool = True,
                            num_parallel_pixels: int = 1000000) -> pd.DataFrame:
        """Assigns SOM clusters to a dataset (reference: cluster_helpers.py:270-301)."""
        external_data_norm = self.normalize_data(external_data) if normalize_data \
            else external_data.copy()
        som_labels = super().generate_som_clusters(
            external_data_norm, num_parallel_obs=num_parallel_pixels
        )

        external_data_norm['pixel_som_cluster'] = som_labels
        self.som_clusters_seen.update(list(np.unique(som_labels)))
        return external_data_norm


class CellSOMCluster(PixieSOMCluster):
    def __init__(self, cell_data: pd.DataFrame, weights_path: pathlib.Path,
                 fovs: List[str], columns: List[str], num_passes: int = 1,
                 xdim: int = 10, ydim: int = 10, lr_start: float = 0.05, lr_end: float = 0.01,
                 seed=42, normalize=True):
        """Cell SOM cluster object (reference: cluster_helpers.py:304-353)."""
        super().__init__(
            weights_path, columns, num_passes, xdim, ydim, lr_start, lr_end, seed
        )

        self.cell_data = cell_data
        self.fovs = fovs

        # subset cell_data on just the FOVs specified
        self.cell_data = self.cell_data[
            self.cell_data['fov'].isin(self.fovs)
        ].reset_index(drop=True)

        # since cell_data is the only dataset, we can just normalize it immediately
        if normalize:
            self.normalize_data()

    def normalize_data(self):
        """99.9 % normalisation of the count columns, zeros ignored
        (reference: cluster_helpers.py:355-372)."""
        cell_data_sub = self.cell_data[self.columns].copy()
        cell_norm_vals = cell_data_sub.replace(0, np.nan).quantile(q=0.999, axis=0)
        cell_data_sub = cell_data_sub.div(cell_norm_vals)
        self.cell_data[self.columns] = cell_data_sub

    def train_som(self, overwrite=False):
        """Trains the SOM using ``cell_data`` (reference: cluster_helpers.py:374-393)."""
        if overwrite:
            warnings.warn('Overwrite flag set, retraining SOM')
        elif self.weights is not None:
            if set(self.weights.columns.values) == set(self.columns):
                warnings.warn('Cell SOM already trained on specified columns')
                return
            warnings.warn('New columns specified, retraining')

        super().train_som(self.cell_data[self.columns])

    def assign_som_clusters(self, num_parallel_cells=1000000) -> pd.DataFrame:
        """Assigns SOM clusters to ``cell_data`` (reference: cluster_helpers.py:395-416)."""
        som_labels = super().generate_som_clusters(
            self.cell_data[self.columns], num_parallel_obs=num_parallel_cells
        )
        self.cell_data['cell_som_cluster'] = som_labels
        return self.cell_data
