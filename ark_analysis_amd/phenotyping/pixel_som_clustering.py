"""Pixel SOM pipeline -- the functions ``ark.phenotyping.pixel_som_clustering`` gives the Pixie notebook
(/root/reference/src/ark/phenotyping/pixel_som_clustering.py): ``train_pixel_som`` (:16-90),
``run_pixel_som_assignment`` (:93-136), ``cluster_pixels`` (:139-289), ``generate_som_avg_files``
(:308-371).  Signatures, defaults, checks, printed lines, restart behaviour and files on disk are the
reference's; the work between them is organised around :mod:`ark_analysis_amd.fov_tables`.

``cluster_pixels`` here is a three-stage pipeline over the FOV tables -- a reader thread fetching the
next tables, the caller's thread normalising + labelling on the GPU, a writer thread storing results in
``<data_dir>_temp`` -- because once the BMU search runs at HBM speed the feather round trip is all that
is left (SURVEY.md section 8 f, rank 1).  ``multiprocess=True`` therefore stays a *hint*: FOVs are still
handled in the calling process (the GPU is the parallel resource), in groups of ``batch_size`` with the
per-group progress lines of the reference's process-pool branch (:257-271).  A side effect worth
knowing: ``som_clusters_seen`` survives in that mode, whereas the reference loses it in the pickled
worker copies.
"""
import os
from typing import Any, Callable, Tuple

from .. import arrow_assign, distributed, flowsom, fov_tables
from ..fov_tables import FovTableDir, TablePrefetcher, TableWriter
from ..host_utils import natsorted, validate_paths, verify_in_list, verify_same_elements
from . import cluster_helpers, pixel_cluster_utils

_CORRUPT = "The data for FOV %s has been corrupted, skipping"


def train_pixel_som(fovs, channels, base_dir,
                    subset_dir='pixel_mat_subsetted',
                    norm_vals_name='post_rowsum_chan_norm.feather',
                    som_weights_name='pixel_som_weights.feather', xdim=10, ydim=10,
                    lr_start=0.05, lr_end=0.01, num_passes=1, seed=42,
                    overwrite=False, *, train_mode="online", batch_steps=None):
    """Train the pixel SOM on the sub-sampled tables of ``base_dir/subset_dir`` and store the codebook
    in ``base_dir/som_weights_name``; returns the :class:`~.cluster_helpers.PixelSOMCluster`.

    Beyond the reference (keyword-only): ``train_mode="batch"`` selects the data-parallel batch rule (throughput mode;
    under ``torchrun`` the training tables are sharded by rank and the per-step statistics all-reduced) on the schedule
    ``batch_steps`` names: an int = that many equal mini-batch steps per pass, ``None`` / "two-phase" = the default
    schedule (``ark_analysis_amd.schedule``).  The default mode is the reference's online rule."""
    distributed.init_from_env()
    subset_root = os.path.join(base_dir, subset_dir)
    norm_file = os.path.join(base_dir, norm_vals_name)
    validate_paths([subset_root, norm_file])   # the weights file may legitimately not exist yet

    subset = FovTableDir(subset_root)
    verify_in_list(provided_fovs=fovs, subsetted_fovs=subset.fovs())
    first_table = fov_tables.read_dataframe(os.path.join(subset_root, subset.files()[0]))
    verify_in_list(provided_channels=channels, subsetted_channels=first_table.columns.values)

    som = cluster_helpers.PixelSOMCluster(
        subset_root, norm_file, os.path.join(base_dir, som_weights_name), fovs, channels,
        num_passes=num_passes, xdim=xdim, ydim=ydim, lr_start=lr_start, lr_end=lr_end, seed=seed,
        train_mode=train_mode, batch_steps=batch_steps)
    _say("Training SOM")
    som.train_som(overwrite=overwrite)
    return som


def _say(*args) -> None:
    """The reference's progress lines, once per job: rank 0 prints, the other ranks stay silent."""
    if distributed.context()[0] == 0:
        print(*args)


_DEVICE_BMU = flowsom.map_data_to_nodes   # the real device entry point (tests may swap the attribute)


def _label_table(som, table, relabel: bool, block: int, host_blocks=None):
    """One FOV table -> the same table with normalised channels and ``pixel_som_cluster``.
    ``relabel``: the table was produced by an earlier run (already normalised, old labels present).
    Arrow tables take the pandas-free device path when it covers them (and the device entry point has
    not been replaced); everything else goes through ``PixelSOMCluster.assign_som_clusters``.
    ``host_blocks`` (an ``arrow_assign.HostBlocks``): the result is ``(table, release-or-None, totals-or-None)``."""
    if host_blocks is not None:
        if (not hasattr(table, "iloc") and flowsom.map_data_to_nodes is _DEVICE_BMU
                and arrow_assign.applicable(som, table, not relabel)):
            return arrow_assign.label_table(som, table, normalize=not relabel, blocks=host_blocks)
        return _label_table(som, table, relabel, block), None, None
    if not hasattr(table, "iloc"):   # an Arrow table
        if flowsom.map_data_to_nodes is _DEVICE_BMU and arrow_assign.applicable(som, table, not relabel):
            return arrow_assign.label_table(som, table, normalize=not relabel)
        table = table.to_pandas()
    if relabel:
        table = table.drop(columns="pixel_som_cluster", errors="ignore")
    return som.assign_som_clusters(table, normalize_data=not relabel, num_parallel_pixels=block)


def _after_write(release, cache, fov, staged_path, totals):
    """Writer-thread epilogue of one table: hand the host block back, remember the table's totals."""
    def done():
        if release is not None:
            release()
        if totals is not None:
            cache.remember(fov, staged_path, totals)
    return done


def run_pixel_som_assignment(pixel_data_path, pixel_pysom_obj, overwrite, num_parallel_pixels, fov):
    """Label one FOV: read ``<pixel_data_path>/<fov>.feather``, write the labelled table to
    ``<pixel_data_path>_temp``.  Returns ``(fov, 0)``, or ``(fov, 1)`` if the table cannot be read."""
    tables = FovTableDir(pixel_data_path)
    try:
        table = tables.load(fov)
    except fov_tables.UNREADABLE:
        return fov, 1
    labelled = _label_table(pixel_pysom_obj, table, overwrite, num_parallel_pixels)
    fov_tables.write_dataframe(labelled, tables.path(fov, staged=True), compression='uncompressed')
    return fov, 0


def _check_columns_against(som, tables: FovTableDir) -> None:
    """The first readable table must carry exactly the channels of the norm row and of the codebook,
    in the same order."""
    probe = tables.first_readable()
    if probe is None:
        raise FileNotFoundError("no readable FOV table in %s" % tables.root)
    probe = fov_tables.unify_label_column(probe)
    channels = fov_tables.feature_columns(probe).values
    verify_same_elements(enforce_order=True, norm_vals_columns=som.norm_data.columns.values,
                         pixel_data_columns=channels)
    verify_same_elements(enforce_order=True, pixel_som_weights_columns=som.weights.columns.values,
                         pixel_data_columns=channels)


def cluster_pixels(fovs, base_dir, pixel_pysom, data_dir='pixel_mat_data',
                   multiprocess=False, batch_size=5, num_parallel_pixels=1000000,
                   overwrite=False):
    """Give every pixel of every FOV table in ``base_dir/data_dir`` its SOM cluster and rewrite the
    tables (channels normalised, ``pixel_som_cluster`` added).  Restartable: tables already staged in
    ``<data_dir>_temp`` are not redone unless ``overwrite``."""
    root = os.path.join(base_dir, data_dir)
    validate_paths([root])
    if pixel_pysom.weights is None:
        raise ValueError("Using untrained pixel_pysom object, please invoke train_pixel_som first")

    tables = FovTableDir(root)
    verify_in_list(provided_fovs=fovs, subsetted_fovs=tables.fovs())
    _check_columns_against(pixel_pysom, tables)

    rank, world = distributed.init_from_env()
    if overwrite:
        _say('Overwrite flag set, reassigning SOM cluster labels to all FOVs')
        pixel_pysom.som_clusters_seen = set()
        if rank == 0:
            tables.open_staging()
        todo = tables.fovs()
    else:
        # (creates the staging directory on a fresh run: one rank decides, everybody uses its answer)
        todo = pixel_cluster_utils.find_fovs_missing_col(base_dir, data_dir, 'pixel_som_cluster') if rank == 0 else None
        todo = distributed.broadcast_object(todo, 0)
    distributed.barrier()
    todo = natsorted(set(todo).intersection(fovs))   # one order on every rank: FOVs are dealt round robin

    if not todo:
        _say("There are no more FOVs to assign SOM labels to, skipping")
        return
    if len(todo) < len(fovs):
        _say("Restarting SOM label assignment from fov %s, "
             "%d fovs left to process" % (todo[0], len(todo)))
    _say("Mapping pixel data to SOM cluster labels")

    # progress is reported per group: batch_size FOVs when multiprocess, else every 10th FOV + the last
    mine = distributed.shard(todo, rank, world)
    group = batch_size if multiprocess else 1
    done = 0
    # FOV tables are independent: read and written side by side.  Per 218 MB table: ~20 ms to read on a reader
    # thread, ~8-14 ms on this thread (copies both ways + the kernels), ~30 ms to serialise on a writer thread.
    # (Two labelling threads with a HIP stream each -- one table's download under the next one's upload -- were
    # measured too: 17.9 instead of 20.0 ms per table on 40 tables, slower on 6; not kept.)
    writer = TableWriter(depth=8, workers=6)
    feed = TablePrefetcher(tables, mine, depth=4, as_arrow=True, workers=3)
    host_blocks = arrow_assign.HostBlocks()
    # per-cluster totals of every table labelled here, keyed by FOV and stamped with the written file's size and
    # mtime: generate_som_avg_files then needs no second pass over 218 MB per FOV (pixel_cluster_utils._CachedTotals)
    cache = pixel_cluster_utils._CachedTotals.attach(pixel_pysom, root)
    try:
        rows = iter(feed)
        for names in fov_tables.batches(mine, group):
            spoiled = []
            for _ in names:
                fov, table = next(rows)
                if table is None:
                    spoiled.append(fov)
                    continue
                labelled, release, totals = _label_table(pixel_pysom, table, overwrite, num_parallel_pixels, host_blocks)
                writer.submit(labelled, tables.path(fov, staged=True),
                              done=_after_write(release, cache, fov, tables.path(fov, staged=True), totals))
            for fov in spoiled:
                print(_CORRUPT % fov)
            done += len(names) - len(spoiled)
            if world == 1 and (multiprocess or done % 10 == 0 or done == len(todo)):
                print("Processed %d fovs" % done)
    finally:
        feed.close()
        writer.close()
        host_blocks.close()

    if world > 1:
        # FOV files were dealt round robin: what the ranks saw is united, then one rank swaps the directories
        seen = set()
        for part in distributed.allgather_objects(sorted(pixel_pysom.som_clusters_seen)):
            seen.update(part)
        pixel_pysom.som_clusters_seen = seen
        total = sum(distributed.allgather_objects(done))
        _say("Processed %d fovs" % total)
    if rank == 0:
        tables.commit(on_rm_error=_ignore_extended_attributes)
        fov_tables.wait_for_cleanup()      # nobody lists or deletes base_dir under the cleaner (reference: a blocking rmtree)
    distributed.barrier()


def _ignore_extended_attributes(func: Callable, filename: str, exc_info: Tuple[Any, Any, Any]):
    """``shutil.rmtree`` error hook: macOS "._*" companion files may refuse deletion; anything else is
    a real error."""
    if func is os.unlink and os.path.basename(filename).startswith("._"):
        return
    raise


def generate_som_avg_files(fovs, channels, base_dir, pixel_pysom, data_dir='pixel_data_dir',
                           pc_chan_avg_som_cluster_name='pixel_channel_avg_som_cluster.csv',
                           num_fovs_subset=100, require_all_som_clusters=True, seed=42,
                           overwrite=False):
    """Write the per-SOM-cluster mean channel expression table (with pixel counts) as CSV to
    ``base_dir/pc_chan_avg_som_cluster_name``."""
    target = os.path.join(base_dir, pc_chan_avg_som_cluster_name)
    if pixel_pysom.weights is None:
        raise ValueError("Using untrained pixel_pysom object, please invoke train_som first")

    rank, _ = distributed.init_from_env()
    exists = distributed.broadcast_object(os.path.exists(target) if rank == 0 else None, 0)
    if exists:
        if not overwrite:
            _say("Already generated SOM cluster channel average file, skipping")
            return
        _say("Overwrite flag set, regenerating SOM cluster channel average file")

    _say("Computing average channel expression across pixel SOM clusters")
    expected = len(pixel_pysom.som_clusters_seen) if require_all_som_clusters else None
    # (with a process group the chosen FOV files are dealt to the ranks and the totals all-reduced)
    means = pixel_cluster_utils.compute_pixel_cluster_channel_avg(
        fovs, channels, base_dir, 'pixel_som_cluster', expected, data_dir,
        num_fovs_subset=num_fovs_subset, seed=seed, keep_count=True,
        _cached=pixel_cluster_utils._CachedTotals.of(pixel_pysom, os.path.join(base_dir, data_dir)))
    if rank == 0:
        means.to_csv(target, index=False)
    distributed.barrier()
