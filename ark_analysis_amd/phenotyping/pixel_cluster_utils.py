"""The parts of ``ark.phenotyping.pixel_cluster_utils`` that sit on the Pixie SOM path
(/root/reference/src/ark/phenotyping/pixel_cluster_utils.py): row normalisation (:109-142), the
per-cluster mean-expression table (:294-416), the restart helper (:419-478) and the TIFF-side percentiles
that feed ``create_pixel_matrix`` (:16-106, :145-181; numpy's float32 arithmetic reproduced on the device).  Same names, arguments,
error / warning texts and results; the TIFF-bound helpers of that module are out of scope (SURVEY.md
section 8).  The per-cluster reduction is one accumulating device pass per FOV (pxsom_cluster_sums:
binary64 sums, int64 counts) instead of a pandas groupby per FOV plus a groupby over the concatenation.
"""
import os
import random
import warnings

import numpy as np
import pandas as pd

from .. import flowsom, image_io
from ..fov_tables import FovTableDir, TablePrefetcher
from ..host_utils import natsort_key, validate_paths, verify_in_list


def _gathered_by_fov(fovs, ranks, per_fov_values):
    """``per_fov_values(my_fovs) -> list`` evaluated on this rank's share of ``fovs`` and put back together in the
    order of ``fovs`` on every rank (``ranks = (rank, world)``; None or one rank: everything here)."""
    from .. import distributed
    if ranks is None or ranks[1] <= 1:
        return per_fov_values(list(fovs))
    mine = distributed.shard(fovs, *ranks)
    known = {}
    for part in distributed.allgather_objects(dict(zip(mine, per_fov_values(mine)))):
        known.update(part)
    return [known[fov] for fov in fovs]


def calculate_channel_percentiles(tiff_dir, fovs, channels, img_sub_folder, percentile, stacks=None, ranks=None):
    """One normalisation value per channel: the ``percentile`` quantile of the positive pixels of each
    FOV's channel image, averaged over the FOVs that have any (reference: pixel_cluster_utils.py:16-58).
    Returns a one-row DataFrame, columns naturally sorted.  ``stacks``: an ``image_io.stack_cache()`` shared
    with the other passes over the same TIFFs.  ``ranks`` = (rank, world) under a process group: every rank
    takes its share of the FOVs, the per-FOV values are gathered and averaged in the order of ``fovs`` -- the
    same numbers on every rank, whatever the rank count."""
    # one device call per FOV covers all its channels (the reference walks channel by channel; the values it
    # averages -- and their order, FOV by FOV -- are the same); the next FOV is decoded meanwhile
    by_fov = _gathered_by_fov(fovs, ranks, lambda mine: [
        flowsom.positive_quantile_f32(stack, percentile)
        for _, stack in image_io.iter_stacks(tiff_dir, mine, channels, img_sub_folder, cache=stacks)])
    per_channel = []
    for j in range(len(channels)):
        found = [values[j] for values in by_fov if not np.isnan(values[j])]   # no positive pixel: not counted
        per_channel.append(np.mean(found))
    table = pd.DataFrame(np.expand_dims(per_channel, axis=0), columns=channels)
    return table[sorted(table.columns, key=natsort_key)]


def calculate_pixel_intensity_percentile(tiff_dir, fovs, channels, img_sub_folder, channel_percentiles,
                                         percentile=0.05, stacks=None, ranks=None):
    """Mean over FOVs of the ``percentile`` quantile of the per-pixel total signal, each channel first
    divided by its normalisation value (reference: pixel_cluster_utils.py:61-106).  ``ranks``: as above."""
    divisors = channel_percentiles.iloc[0].values
    per_fov = _gathered_by_fov(fovs, ranks, lambda mine: [
        flowsom.total_intensity_quantile_f32(stack, divisors, percentile)
        for _, stack in image_io.iter_stacks(tiff_dir, mine, channels, img_sub_folder, cache=stacks)])
    return np.mean(per_fov)


def check_for_modified_channels(tiff_dir, test_fov, img_sub_folder, channels):
    """Warn when a selected channel also exists in a post-processed variant (``_smoothed``,
    ``_nuc_include``, ``_nuc_exclude``) the user may have meant (reference: pixel_cluster_utils.py:145-181)."""
    present = set(image_io.channel_names(tiff_dir, test_fov, img_sub_folder))
    for channel in channels:
        for suffix in ('_smoothed', '_nuc_include', '_nuc_exclude'):
            variant = channel + suffix
            if variant in present:
                warnings.warn('You selected {} as the channel to analyze, but there were potential'
                              ' modified channels found: {}. Make sure you selected the correct '
                              'version of the channel for inclusion in '
                              'clustering'.format(channel, variant))


def normalize_rows(pixel_data, channels, include_seg_label=True):
    """Each pixel's channel values divided by their sum; position (and label) columns carried along."""
    values = pixel_data[channels]
    out = values.div(values.sum(axis=1), axis=0)
    carried = ["fov", "row_index", "column_index"] + (["label"] if include_seg_label else [])
    out[carried] = pixel_data.loc[out.index.values, carried]
    return out


_DEVICE_SUMS = flowsom.cluster_sums   # the real device entry point (tests may swap the attribute)


class _ClusterTotals:
    """Running per-cluster channel sums and pixel counts over FOV tables; cluster ids are whatever the
    label column holds (SOM labels 1..K, or meta-cluster ids), so totals are kept per id."""

    def __init__(self, channels):
        self.channels = list(channels)
        self._sum = {}
        self._n = {}

    def add_table(self, table: pd.DataFrame, cluster_col: str) -> None:
        ids = table[cluster_col].to_numpy()
        if ids.size == 0:
            return
        distinct, dense = np.unique(ids, return_inverse=True)
        # the kernel wants labels 1..len(distinct)
        sums, counts = flowsom.cluster_sums(table[self.channels].to_numpy(),
                                            (dense + 1).astype(np.int32), len(distinct))
        self._merge(distinct, sums, counts)

    def _merge(self, distinct, sums, counts) -> None:
        for row, cid in enumerate(distinct):
            if cid in self._sum:
                self._sum[cid] = self._sum[cid] + sums[row]
                self._n[cid] += int(counts[row])
            else:
                self._sum[cid] = np.array(sums[row], dtype=np.float64)
                self._n[cid] = int(counts[row])

    def add_arrow(self, table, cluster_col: str) -> None:
        """The same totals from the Arrow table the file reader produced, without pandas: channel chunks ->
        ``[C, n]`` in HBM -> transpose -> ``pxsom_cluster_sums`` with the distinct ids found on the device.
        Tables it does not cover (nulls, other column types) take :meth:`add_table`."""
        import pyarrow as pa
        import torch
        from .. import _capi, som_device
        ids_col = table.column(cluster_col) if cluster_col in table.column_names else None
        plain = (ids_col is not None and ids_col.null_count == 0 and table.num_rows > 0
                 and (pa.types.is_integer(ids_col.type) or pa.types.is_floating(ids_col.type))
                 and all(ch in table.column_names and table.column(ch).type == pa.float64()
                         and table.column(ch).null_count == 0 for ch in self.channels))
        if not plain:
            return self.add_table(table.to_pandas(), cluster_col)
        dev = _capi.require_gpu()
        n, c = table.num_rows, len(self.channels)
        planar = torch.empty((c, n), dtype=torch.float64, device=dev)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", UserWarning)   # torch: "array is not writable" (only read)
            for j, name in enumerate(self.channels):
                at = 0
                for chunk in table.column(name).chunks:
                    host = chunk.to_numpy(zero_copy_only=True)
                    planar[j, at:at + len(host)].copy_(torch.from_numpy(host), non_blocking=True)
                    at += len(host)
            ids = torch.from_numpy(ids_col.to_numpy()).to(dev)
        distinct, dense = torch.unique(ids, return_inverse=True)
        sums, counts = som_device.cluster_sums(planar.t().contiguous(), (dense + 1).to(torch.int32), int(distinct.numel()))
        self._merge(distinct.cpu().numpy(), sums.cpu().numpy(), counts.cpu().numpy())

    def merge_ranks(self) -> None:
        """Totals of every rank united on every rank (one gather of the small per-cluster tables; ids are
        dictionary keys, so ranks that met different clusters merge cleanly).  Summation in rank order."""
        from .. import distributed
        if distributed.context()[1] <= 1:
            return
        parts = distributed.allgather_objects((self._sum, self._n))
        self._sum, self._n = {}, {}
        for sums, counts in parts:
            for cid in sums:
                if cid in self._sum:
                    self._sum[cid] = self._sum[cid] + sums[cid]
                    self._n[cid] += counts[cid]
                else:
                    self._sum[cid] = np.array(sums[cid], dtype=np.float64)
                    self._n[cid] = int(counts[cid])

    def __len__(self):
        return len(self._sum)

    def frame(self, cluster_col: str) -> pd.DataFrame:
        """[cluster id | channel sums ... | count], ascending cluster id."""
        order = sorted(self._sum)
        out = pd.DataFrame(np.stack([self._sum[cid] for cid in order]), columns=self.channels)
        out.insert(0, cluster_col, order)
        out["count"] = [self._n[cid] for cid in order]
        return out


class _CachedTotals:
    """Per-cluster channel sums / pixel counts of the FOV tables ``cluster_pixels`` has just written, kept on the SOM
    object: the rows were in HBM when they were labelled, so their table cost one 20 us kernel -- re-reading 218 MB
    per FOV for it (what the reference's generate_som_avg_files does) costs a thousand times that.  An entry is
    used only while the file it describes is still the one on disk (same size and modification time)."""

    def __init__(self, root: str):
        self.root = os.path.abspath(root)
        self._entries = {}

    @classmethod
    def attach(cls, som, root: str) -> "_CachedTotals":
        cache = cls(root)
        som._fov_totals = cache
        return cache

    @staticmethod
    def of(som, root: str):
        cache = getattr(som, "_fov_totals", None)
        return cache if cache is not None and cache.root == os.path.abspath(root) else None

    def remember(self, fov: str, written_path: str, totals) -> None:
        try:
            st = os.stat(written_path)      # the staging file: the directory swap keeps size and mtime
        except OSError:
            return
        self._entries[fov] = (st.st_size, st.st_mtime_ns, totals)

    def lookup(self, fov: str, channels):
        entry = self._entries.get(fov)
        if entry is None:
            return None
        try:
            st = os.stat(os.path.join(self.root, fov + ".feather"))
        except OSError:
            return None
        size, mtime_ns, (feats, sums, counts) = entry
        if (st.st_size, st.st_mtime_ns) != (size, mtime_ns) or any(ch not in feats for ch in channels):
            return None
        cols = [feats.index(ch) for ch in channels]
        present = np.flatnonzero(counts)
        return present + 1, sums[np.ix_(present, cols)], counts[present]


def compute_pixel_cluster_channel_avg(fovs, channels, base_dir, pixel_cluster_col,
                                      num_pixel_clusters,
                                      pixel_data_dir='pixel_mat_data',
                                      num_fovs_subset=100, seed=42, keep_count=False, *, _cached=None):
    """Mean channel expression of every pixel SOM / meta cluster over (a random subset of) the FOVs.

    ``num_pixel_clusters``: how many clusters the table must contain (``None``: do not check);
    ``num_fovs_subset``: how many FOVs to draw (``random.seed(seed)``; all of them if fewer exist);
    ``keep_count``: keep the per-cluster pixel count column."""
    verify_in_list(provided_cluster_col=[pixel_cluster_col],
                   valid_cluster_cols=['pixel_som_cluster', 'pixel_meta_cluster'])
    if num_pixel_clusters is not None and num_pixel_clusters <= 0:
        raise ValueError("If set, number of pixel clusters desired must be a positive integer")
    if num_fovs_subset <= 0:
        raise ValueError("Number of fovs to subset must be a positive integer")

    available = len(fovs)
    if available < num_fovs_subset:
        warnings.warn('Provided num_fovs_subset=%d but only %d FOVs in dataset, subsetting just the %d FOVs'
                      % (num_fovs_subset, available, available))
    random.seed(seed)
    chosen = fovs if num_fovs_subset >= available else random.sample(fovs, num_fovs_subset)

    tables = FovTableDir(os.path.join(base_dir, pixel_data_dir))
    totals = _ClusterTotals(channels)
    # tables are read one ahead on a background thread; with the device entry point in place (tests may swap
    # it) they stay Arrow tables and never become DataFrames.  With a process group every rank takes its share
    # of the chosen files (the same list everywhere: it comes from the seeded draw above).
    from .. import distributed
    on_device = flowsom.cluster_sums is _DEVICE_SUMS
    mine = distributed.shard(chosen)
    if _cached is not None and on_device and pixel_cluster_col == 'pixel_som_cluster':
        # tables this process labelled a moment ago (and that are still the files on disk) are not read again
        unread = []
        for fov in mine:
            hit = _cached.lookup(fov, totals.channels)
            if hit is None:
                unread.append(fov)
            else:
                totals._merge(*hit)
        mine = unread
    feed = TablePrefetcher(tables, mine, depth=4, as_arrow=on_device, workers=3)
    try:
        for fov, table in feed:
            if table is None:
                print("The data for FOV %s has been corrupted, skipping" % fov)
            elif on_device:
                totals.add_arrow(table, pixel_cluster_col)
            else:
                totals.add_table(table, pixel_cluster_col)
    finally:
        feed.close()
    totals.merge_ranks()

    if len(totals) == 0:
        raise ValueError("No objects to concatenate")   # what pd.concat([]) says in the reference
    table = totals.frame(pixel_cluster_col)
    if num_pixel_clusters is not None and len(table) < num_pixel_clusters:
        raise ValueError('Averaged data contains just %d clusters out of %d. '
                         'Average expression file not written. '
                         'Consider increasing your num_fovs_subset value.' % (len(table), num_pixel_clusters))

    table[totals.channels] = table[totals.channels].div(table['count'], axis=0)
    table[pixel_cluster_col] = table[pixel_cluster_col].astype(int)
    table = table.sort_values(by=pixel_cluster_col)
    return table if keep_count else table.drop('count', axis=1)


def find_fovs_missing_col(base_dir, data_dir, missing_col):
    """FOVs of ``base_dir/data_dir`` a stage adding ``missing_col`` still has to process; creates
    ``<data_dir>_temp`` when it starts a fresh run (see :meth:`FovTableDir.pending`)."""
    root = os.path.join(base_dir, data_dir)
    validate_paths(root)
    return FovTableDir(root).pending(missing_col)
