"""Pixel meta-clustering -- the functions ``ark.phenotyping.pixel_meta_clustering`` gives the Pixie notebook
(/root/reference/src/ark/phenotyping/pixel_meta_clustering.py): ``run_pixel_consensus_assignment`` (:17-50),
``pixel_consensus_cluster`` (:53-188), ``generate_meta_avg_files`` (:191-275), ``update_pixel_meta_labels``
(:278-330), ``apply_pixel_meta_cluster_remapping`` (:333-446), ``generate_remap_avg_files`` (:449-534).
Signatures, defaults, checks, printed lines, restart behaviour and files on disk are the reference's.

What the step is, numerically: a K-row table (K SOM clusters x channel means, written by
``generate_som_avg_files``) is z-scored, capped and cut into ``max_k`` groups by Ward agglomeration on the host
(``cluster_helpers.PixieConsensusCluster``; microseconds to milliseconds of work); what touches the per-pixel data
is a K-entry lookup table.  The FOV tables are walked the way ``cluster_pixels`` walks them -- read ahead on one
thread, relabel, write behind on others -- and the relabelling itself works on the Arrow column the file reader
produced (``_relabel_arrow``: one ``take`` through the lookup table, untouched columns passed through zero-copy);
tables that path does not cover (labels missing from the mapping, nulls, non-integer label columns) go through
pandas exactly as the reference does.  With a process group the FOV files are dealt to the ranks.
"""
import json
import os
from typing import Dict, Optional, Sequence

import numpy as np
import pandas as pd
import pyarrow as pa

from .. import distributed, fov_tables
from ..fov_tables import FovTableDir, TablePrefetcher, TableWriter
from ..host_utils import natsorted, validate_paths, verify_in_list
from . import cluster_helpers, pixel_cluster_utils

_CORRUPT = "The data for FOV %s has been corrupted, skipping"
SOM_COL, META_COL, RENAME_COL = 'pixel_som_cluster', 'pixel_meta_cluster', 'pixel_meta_cluster_rename'


def _say(*args) -> None:
    """Progress lines once per job: rank 0 prints, the other ranks stay silent."""
    if distributed.context()[0] == 0:
        print(*args)


# ---- Arrow-level relabelling ------------------------------------------------------------------------------------
def _pandas_field(name: str, sample: pd.Series) -> dict:
    """The pandas-metadata entry ``pa.Table.from_pandas`` writes for a column like ``sample``."""
    probe = pa.Table.from_pandas(pd.DataFrame({name: sample}), preserve_index=None)
    meta = json.loads(probe.schema.metadata[b"pandas"].decode())
    return next(c for c in meta["columns"] if c["name"] == name)


def _with_columns(table: pa.Table, new: Dict[str, pa.Array], samples: Dict[str, pd.Series]) -> pa.Table:
    """``table`` with the columns of ``new`` replaced in place (if present) or appended (in the order given), and the
    pandas metadata brought in line -- the table ``DataFrame.__setitem__`` + ``write_dataframe`` would produce."""
    names = [n for n in table.column_names]
    columns = [table.column(n) for n in names]
    for name, arr in new.items():
        if name in names:
            columns[names.index(name)] = arr
        else:
            names.append(name)
            columns.append(arr)
    out = pa.Table.from_arrays(columns, names=names)
    schema_meta = table.schema.metadata
    if schema_meta and b"pandas" in schema_meta:
        meta = json.loads(schema_meta[b"pandas"].decode())
        known = {c["name"]: c for c in meta.get("columns", [])}
        for name in new:
            known[name] = _pandas_field(name, samples[name])
        rest = [c for c in meta.get("columns", []) if c["name"] not in names]     # index entries
        meta["columns"] = [known[n] for n in names if n in known] + rest
        schema_meta = dict(schema_meta)
        schema_meta[b"pandas"] = json.dumps(meta).encode()
    return out.replace_schema_metadata(schema_meta)


def _int_labels(table: pa.Table, column: str) -> Optional[np.ndarray]:
    """The integer label column as one numpy vector, or ``None`` when the fast path does not apply."""
    if column not in table.column_names:
        return None
    col = table.column(column)
    if not pa.types.is_integer(col.type) or col.null_count:
        return None
    return col.to_numpy()


def _take(lut: np.ndarray, labels: np.ndarray) -> Optional[np.ndarray]:
    """``lut[labels]``, or ``None`` if a label has no entry (``Series.map`` would give NaN: pandas path)."""
    if labels.size == 0:
        return lut[:0].copy()
    if labels.min() < 0 or labels.max() >= lut.size:
        return None
    got = lut[labels]
    return None if (got < 0).any() else got


def _relabel_arrow(table: pa.Table, lut: np.ndarray) -> Optional[pa.Table]:
    """``assign_consensus_labels`` on an Arrow table: ``pixel_meta_cluster`` = lut[pixel_som_cluster]."""
    labels = _int_labels(table, SOM_COL)
    meta = None if labels is None else _take(lut, labels)
    if meta is None:
        return None
    return _with_columns(table, {META_COL: pa.array(meta)}, {META_COL: pd.Series(meta[:1])})


# ---- the shared walk over the FOV tables --------------------------------------------------------------------------
def _rewrite_tables(tables: FovTableDir, todo: Sequence[str], relabel, multiprocess: bool, batch_size: int) -> None:
    """Every table of ``todo`` through ``relabel(table) -> table`` into the staging directory, then the directory
    swap: reader thread -> caller's thread -> writer threads; FOVs dealt round robin over the ranks; the
    reference's progress lines (per ``batch_size`` group when ``multiprocess``, else every 10th FOV and the last)."""
    rank, world = distributed.context()
    mine = distributed.shard(todo, rank, world)
    group = batch_size if multiprocess else 1
    done = 0
    writer = TableWriter(depth=8, workers=6)
    feed = TablePrefetcher(tables, mine, depth=4, as_arrow=True, workers=3)
    try:
        rows = iter(feed)
        for names in fov_tables.batches(mine, group):
            spoiled = []
            for _ in names:
                fov, table = next(rows)
                if table is None:
                    spoiled.append(fov)
                    continue
                writer.submit(relabel(table), tables.path(fov, staged=True))
            for fov in spoiled:
                print(_CORRUPT % fov)
            done += len(names) - len(spoiled)
            if world == 1 and (multiprocess or done % 10 == 0 or done == len(todo)):
                print("Processed %d fovs" % done)
    finally:
        feed.close()
        writer.close()
    if world > 1:
        _say("Processed %d fovs" % sum(distributed.allgather_objects(done)))
    distributed.barrier()
    if rank == 0:
        tables.commit()
        fov_tables.wait_for_cleanup()
    distributed.barrier()


# ---- consensus clustering -------------------------------------------------------------------------------------------
def run_pixel_consensus_assignment(pixel_data_path, pixel_cc_obj, fov):
    """Meta-label one FOV: read ``<pixel_data_path>/<fov>.feather``, write the table with ``pixel_meta_cluster`` to
    ``<pixel_data_path>_temp``.  Returns ``(fov, 0)``, or ``(fov, 1)`` if the table cannot be read."""
    tables = FovTableDir(pixel_data_path)
    try:
        table = tables.load(fov)
    except fov_tables.UNREADABLE:
        return fov, 1
    fov_tables.write_dataframe(pixel_cc_obj.assign_consensus_labels(table), tables.path(fov, staged=True),
                               compression='uncompressed')
    return fov, 0


def pixel_consensus_cluster(fovs, channels, base_dir, max_k=20, cap=3,
                            data_dir='pixel_mat_data',
                            pc_chan_avg_som_cluster_name='pixel_channel_avg_som_cluster.csv',
                            multiprocess=False, batch_size=5, seed=42, overwrite=False):
    """Meta-cluster the pixel SOM clusters (``max_k`` groups from the z-scored, ``cap``-clipped average table) and
    give every pixel of every FOV table in ``base_dir/data_dir`` its ``pixel_meta_cluster``.  Restartable like
    ``cluster_pixels``.  Returns the :class:`~.cluster_helpers.PixieConsensusCluster` (``None`` if nothing was left
    to do)."""
    root = os.path.join(base_dir, data_dir)
    avg_path = os.path.join(base_dir, pc_chan_avg_som_cluster_name)
    validate_paths([root, avg_path])

    rank, _ = distributed.init_from_env()
    tables = FovTableDir(root)
    if overwrite:
        _say('Overwrite flag set, reassigning meta cluster labels to all FOVs')
        if rank == 0:
            tables.open_staging()
        todo = tables.fovs()
    else:
        todo = pixel_cluster_utils.find_fovs_missing_col(base_dir, data_dir, META_COL) if rank == 0 else None
        todo = distributed.broadcast_object(todo, 0)
    distributed.barrier()
    todo = natsorted(set(todo).intersection(fovs))

    if not todo:
        _say("There are no more FOVs to assign meta labels to, skipping")
        return
    if len(todo) < len(fovs):
        _say("Restarting meta cluster label assignment from fov %s, "
             "%d fovs left to process" % (todo[0], len(todo)))

    pixel_cc = cluster_helpers.PixieConsensusCluster('pixel', avg_path, channels, max_k=max_k, cap=cap)
    _say("z-score scaling and capping data")
    pixel_cc.scale_data()
    np.random.seed(seed)
    _say("Running consensus clustering")
    pixel_cc.run_consensus_clustering()
    pixel_cc.generate_som_to_meta_map()

    _say("Mapping pixel data to consensus cluster labels")
    lut = pixel_cc.lookup_table()

    def relabel(table):
        fast = _relabel_arrow(table, lut)
        return fast if fast is not None else pixel_cc.assign_consensus_labels(table.to_pandas())

    _rewrite_tables(tables, todo, relabel, multiprocess, batch_size)
    return pixel_cc


def generate_meta_avg_files(fovs, channels, base_dir, pixel_cc, data_dir='pixel_mat_data',
                            pc_chan_avg_som_cluster_name='pixel_channel_avg_som_cluster.csv',
                            pc_chan_avg_meta_cluster_name='pixel_channel_avg_meta_cluster.csv',
                            num_fovs_subset=100, seed=42, overwrite=False):
    """Write the per-meta-cluster mean channel expression table (with pixel counts) and add the meta cluster of
    every SOM cluster to the per-SOM-cluster table."""
    som_avg_path = os.path.join(base_dir, pc_chan_avg_som_cluster_name)
    meta_avg_path = os.path.join(base_dir, pc_chan_avg_meta_cluster_name)
    validate_paths([som_avg_path])

    rank, _ = distributed.init_from_env()
    exists = distributed.broadcast_object(os.path.exists(meta_avg_path) if rank == 0 else None, 0)
    if exists:
        if not overwrite:
            _say("Already generated meta cluster channel average file, skipping")
            return
        _say("Overwrite flag set, regenerating meta cluster channel average file")

    _say("Computing average channel expression across pixel meta clusters")
    meta_avg = pixel_cluster_utils.compute_pixel_cluster_channel_avg(
        fovs, channels, base_dir, META_COL, pixel_cc.max_k, data_dir,
        num_fovs_subset=num_fovs_subset, seed=seed, keep_count=True)
    _say("Mapping meta cluster values onto average channel expression across pixel SOM clusters")
    if rank == 0:
        meta_avg.to_csv(meta_avg_path, index=False)
        som_avg = pd.read_csv(som_avg_path)
        if META_COL in som_avg.columns.values:     # an earlier run's column (overwrite)
            som_avg = som_avg.drop(columns=META_COL)
        som_avg[SOM_COL] = som_avg[SOM_COL].astype(int)
        pd.merge_asof(som_avg, pixel_cc.mapping, on=SOM_COL).to_csv(som_avg_path, index=False)
    distributed.barrier()


# ---- manual remapping -------------------------------------------------------------------------------------------------
def update_pixel_meta_labels(pixel_data_path, pixel_remapped_dict, pixel_renamed_meta_dict, fov):
    """Re-label one FOV with the remapping scheme (``pixel_som_cluster`` -> ``pixel_meta_cluster`` ->
    ``pixel_meta_cluster_rename``) into ``<pixel_data_path>_temp``.  ``(fov, 0)`` / ``(fov, 1)`` as above."""
    tables = FovTableDir(pixel_data_path)
    try:
        table = tables.load(fov)
    except fov_tables.UNREADABLE:
        return fov, 1
    fov_tables.write_dataframe(_remap_frame(table, pixel_remapped_dict, pixel_renamed_meta_dict),
                               tables.path(fov, staged=True), compression='uncompressed')
    return fov, 0


def _remap_frame(table: pd.DataFrame, som_to_meta: dict, meta_to_name: dict) -> pd.DataFrame:
    verify_in_list(fov_som_labels=table[SOM_COL].unique(), som_labels_in_mapping=list(som_to_meta.keys()))
    table[META_COL] = table[SOM_COL].map(som_to_meta)
    table[RENAME_COL] = table[META_COL].map(meta_to_name)
    return table


def _remap_arrow(table: pa.Table, som_lut: np.ndarray, name_codes: np.ndarray, names: pa.Array,
                 name_sample: pd.Series) -> Optional[pa.Table]:
    """The same two lookups on the Arrow label column: meta ids by ``take``, names as a dictionary decoded by Arrow."""
    labels = _int_labels(table, SOM_COL)
    meta = None if labels is None else _take(som_lut, labels)
    if meta is None:
        return None
    codes = _take(name_codes, meta)
    if codes is None:
        return None
    renamed = pa.DictionaryArray.from_arrays(pa.array(codes.astype(np.int32)), names).dictionary_decode()
    return _with_columns(table, {META_COL: pa.array(meta), RENAME_COL: renamed},
                         {META_COL: pd.Series(meta[:1]), RENAME_COL: name_sample})


def _dense(mapping: dict) -> Optional[np.ndarray]:
    """An integer -> integer dictionary as a lookup table (-1 = no entry); ``None`` if it is not one."""
    try:
        keys = np.array(list(mapping.keys()))
        vals = np.array(list(mapping.values()))
    except Exception:
        return None
    if keys.size == 0 or keys.dtype.kind not in "iu" or vals.dtype.kind not in "iu" or keys.min() < 0 or vals.min() < 0:
        return None
    lut = np.full(int(keys.max()) + 1, -1, dtype=np.int64)
    lut[keys] = vals
    return lut


def apply_pixel_meta_cluster_remapping(fovs, channels, base_dir, pixel_data_dir, pixel_remapped_name,
                                       multiprocess=False, batch_size=5):
    """Apply the (manually adjusted) SOM -> meta -> name table in ``base_dir/pixel_remapped_name`` to every FOV table
    of ``base_dir/pixel_data_dir``."""
    root = os.path.join(base_dir, pixel_data_dir)
    remap_path = os.path.join(base_dir, pixel_remapped_name)
    validate_paths([root, remap_path])

    remapped = pd.read_csv(remap_path)
    verify_in_list(required_cols=[SOM_COL, META_COL, RENAME_COL], remapped_data_cols=remapped.columns.values)
    som_to_meta = dict(remapped[[SOM_COL, META_COL]].values)
    cluster_helpers.verify_unique_meta_clusters(remapped, meta_cluster_type="pixel")
    meta_to_name = dict(remapped[[META_COL, RENAME_COL]].drop_duplicates().values)

    rank, _ = distributed.init_from_env()
    tables = FovTableDir(root)
    fresh = distributed.broadcast_object((not os.path.exists(tables.staging)) if rank == 0 else None, 0)
    if fresh:
        if rank == 0:
            tables.open_staging()
        todo = list(fovs)
    else:
        todo = pixel_cluster_utils.find_fovs_missing_col(base_dir, pixel_data_dir, RENAME_COL) if rank == 0 else None
        todo = natsorted(distributed.broadcast_object(todo, 0))
        _say("Restarting meta cluster remapping assignment from %s, "
             "%d fovs left to process" % (todo[0], len(todo)))
    distributed.barrier()

    _say("Using re-mapping scheme to re-label pixel meta clusters")
    # dense forms of the two dictionaries for the Arrow path (integer SOM / meta ids; names of any one type)
    som_lut = _dense(som_to_meta)
    name_codes = names = name_sample = None
    if som_lut is not None and all(isinstance(k, (int, np.integer)) and k >= 0 for k in meta_to_name):
        keys = list(meta_to_name.keys())
        name_codes = np.full(int(max(keys)) + 1, -1, dtype=np.int64)
        name_codes[keys] = np.arange(len(keys))
        try:
            name_values = pd.Series(list(meta_to_name.values()))
            names = pa.Array.from_pandas(name_values)
            name_sample = name_values.iloc[:1]
        except Exception:
            names = None

    def relabel(table):
        fast = _remap_arrow(table, som_lut, name_codes, names, name_sample) if names is not None else None
        return fast if fast is not None else _remap_frame(table.to_pandas(), som_to_meta, meta_to_name)

    _rewrite_tables(tables, todo, relabel, multiprocess, batch_size)


def generate_remap_avg_files(fovs, channels, base_dir, pixel_data_dir, pixel_remapped_name,
                             pc_chan_avg_som_cluster_name, pc_chan_avg_meta_cluster_name,
                             num_fovs_subset=100, seed=42):
    """After a remapping: recompute the per-meta-cluster average table (with the renamed column) and re-assign the
    meta columns of the per-SOM-cluster average table."""
    remap_path = os.path.join(base_dir, pixel_remapped_name)
    som_avg_path = os.path.join(base_dir, pc_chan_avg_som_cluster_name)
    meta_avg_path = os.path.join(base_dir, pc_chan_avg_meta_cluster_name)
    validate_paths([remap_path, som_avg_path, meta_avg_path])

    remapped = pd.read_csv(remap_path)
    som_to_meta = dict(remapped[[SOM_COL, META_COL]].values)
    meta_to_name = dict(remapped[[META_COL, RENAME_COL]].drop_duplicates().values)

    rank, _ = distributed.init_from_env()
    _say("Re-computing average channel expression across pixel meta clusters")
    meta_avg = pixel_cluster_utils.compute_pixel_cluster_channel_avg(
        fovs, channels, base_dir, META_COL, len(remapped[META_COL].unique()), pixel_data_dir,
        num_fovs_subset=num_fovs_subset, seed=seed, keep_count=True)
    meta_avg[RENAME_COL] = meta_avg[META_COL].map(meta_to_name)
    _say("Re-assigning meta cluster column in pixel SOM cluster average channel expression table")
    if rank == 0:
        meta_avg.to_csv(meta_avg_path, index=False)
        som_avg = pd.read_csv(som_avg_path)
        som_avg[META_COL] = som_avg[SOM_COL].map(som_to_meta)
        som_avg[RENAME_COL] = som_avg[META_COL].map(meta_to_name)
        som_avg.to_csv(som_avg_path, index=False)
    distributed.barrier()
