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

from .. import distributed, flowsom, image_io
from ..arrow_assign import HostBlocks
from ..fov_tables import FovTableDir, TableWriter, read_dataframe, write_dataframe
from ..host_utils import natsort_key, validate_paths, verify_in_list
from . import pixel_cluster_utils


def _device_rows(channels, img_data, pixel_thresh_val, blur_factor, nonzero_q=None, blocks=None):
    """The device half of create_fov_pixel_data: ``(values [n, C], flat pixel numbers of the kept rows, image
    width, non-zero quantiles or None[, release])`` -- ``release`` with ``blocks``, see flowsom.fov_pixel_rows.  float32 images (what preprocess_fov passes for float32 TIFFs) keep
    float32 semantics end to end: scipy stores each blur pass as float32, pandas sums and divides the float32
    frame in binary32."""
    channels.sort(key=natsort_key)                       # in place, like the reference (:44)
    extra = {} if nonzero_q is None else {"nonzero_q": nonzero_q}
    if blocks is not None:
        extra["blocks"] = blocks
    got = flowsom.fov_pixel_rows(np.asarray(img_data)[:, :, :len(channels)], blur_factor, pixel_thresh_val, **extra)
    out = got[0], got[1], img_data.shape[1], (got[2] if nonzero_q is not None else None)
    return out if blocks is None else out + (got[-1],)


def _assemble_tables(fov, channels, values, kept_h, width, seg_labels, subset_proportion, seed=None):
    """The host half: the reference's DataFrame (channels, fov, row_index, column_index[, label]) and its
    training sub-sample (global numpy RNG state as the reference, :78; ``seed`` re-seeds it first, the way
    preprocess_fov does before the call)."""
    pixel_mat = pd.DataFrame(values, columns=channels)
    pixel_mat['fov'] = fov
    pixel_mat['row_index'] = (kept_h // width).astype(np.int64)
    pixel_mat['column_index'] = (kept_h % width).astype(np.int64)
    if seg_labels is not None:
        pixel_mat['label'] = np.asarray(seg_labels).flatten()[kept_h]
    if seed is not None:
        np.random.seed(seed)
    return pixel_mat, pixel_mat.sample(frac=subset_proportion)


def _fov_tables(fov, channels, img_data, seg_labels, pixel_thresh_val, blur_factor, subset_proportion,
                nonzero_q=None):
    """create_fov_pixel_data plus, on request, the non-zero quantile of every channel of the full table at
    ``nonzero_q`` (taken on the device while the rows are there; same numbers as quantiling the table)."""
    values, kept_h, width, quantiles = _device_rows(channels, img_data, pixel_thresh_val, blur_factor, nonzero_q)
    full, subset = _assemble_tables(fov, channels, values, kept_h, width, seg_labels, subset_proportion)
    return full, subset, quantiles


def create_fov_pixel_data(fov, channels, img_data, seg_labels, pixel_thresh_val,
                          blur_factor=2, subset_proportion=0.1):
    """Preprocess pixel data for one fov; returns ``(pixel_mat, pixel_mat_subset)`` DataFrames with
    the reference's columns (channels, fov, row_index, column_index[, label])."""
    full, subset, _ = _fov_tables(fov, channels, img_data, seg_labels, pixel_thresh_val, blur_factor,
                                  subset_proportion)
    return full, subset


# ---- cohort level: TIFF folders -> per-FOV pixel tables + normalisation files ----------------------------
# (reference: preprocess_fov, pixie_preprocessing.py:83-198, and create_pixel_matrix, :201-456.)

_PER_FOV_QUANTILES = "channel_norm_post_rownorm_perfov.csv"


def _read_segmentation(seg_dir, fov, seg_suffix):
    return image_io.read_image(os.path.join(seg_dir, fov + seg_suffix))


def _fov_device_half(tiff_dir, seg_dir, seg_suffix, img_sub_folder, is_mibitiff, channels, blur_factor,
                     pixel_thresh_val, channel_norm_df, fov, stack=None, post_rownorm_q=None, staging=None,
                     blocks=None):
    """preprocess_fov up to the rows coming back from the device: ``(values, kept pixel numbers, width,
    quantiles or None, segmentation labels or None[, release])``.  ``staging``: a ``flowsom.HostStaging`` that
    receives the divided stack (same numbers; the buffer is reused FOV after FOV and uploads faster); ``blocks``:
    recycled host blocks for the rows (``release()`` returns the block)."""
    if is_mibitiff:
        raise NotImplementedError("multi-page MIBItiff input is not built; export single-channel TIFFs")
    verify_in_list(provided_chans=channels,
                   pixel_mat_chans=image_io.channel_names(tiff_dir, fov, img_sub_folder))
    labels = _read_segmentation(seg_dir, fov, seg_suffix) if seg_dir is not None else None

    if stack is None:
        stack = image_io.read_channels(tiff_dir, fov, channels, img_sub_folder)
    stack = stack.astype(np.float32, copy=False)
    norm = np.array(channel_norm_df.iloc[0].values).reshape([1, 1, -1])
    if staging is not None and norm.dtype == np.float32:
        # the same ufunc call writing into the staging buffer, in the stack's own memory order (channel-planar
        # for stacks from image_io.read_channels)
        h, w, c = stack.shape
        if stack.transpose(2, 0, 1).flags.c_contiguous:
            divided = staging.array((c, h, w), np.float32).transpose(1, 2, 0)
        else:
            divided = staging.array((h, w, c), np.float32)
        stack = np.divide(stack, norm, out=divided)
    else:
        stack = stack / norm                                                       # float32 / float32 stays float32
    q = None if post_rownorm_q is None else (post_rownorm_q * 100) / 100      # pandas' effective q
    got = _device_rows(channels, stack, pixel_thresh_val, blur_factor, nonzero_q=q, blocks=blocks)
    return got[:4] + (labels,) + got[4:]


def _fov_table_half(base_dir, data_dir, subset_dir, channels, subset_proportion, seed, fov, rows, writer=None):
    """The rest of preprocess_fov: seeded DataFrames out of the device's rows, the two writes; returns
    ``(full table, quantile Series or None)``."""
    values, kept_h, width, quantiles, labels = rows[:5]
    full, subset = _assemble_tables(fov, channels, values, kept_h, width, labels, subset_proportion, seed=seed)
    for table, folder in ((full, data_dir), (subset, subset_dir)):
        path = os.path.join(base_dir, folder, fov + ".feather")
        if writer is None:
            write_dataframe(table, path, compression='uncompressed')
        else:
            writer.submit(table, path)
    if quantiles is None:
        return full, None
    return full, pd.Series(quantiles, index=pd.Index(list(channels), name="channel"), name=fov)


def preprocess_fov(base_dir, tiff_dir, data_dir, subset_dir, seg_dir, seg_suffix,
                   img_sub_folder, is_mibitiff, channels, blur_factor,
                   subset_proportion, pixel_thresh_val, seed, channel_norm_df, fov, stack=None,
                   post_rownorm_q=None, writer=None):
    """One FOV from TIFFs to tables: load the channel images, divide by the pre-row-norm channel values,
    run :func:`create_fov_pixel_data` (seeded), write ``<data_dir>/<fov>.feather`` and
    ``<subset_dir>/<fov>.feather``; returns the full table (the caller needs its 99.9 % values).
    Extensions used by :func:`create_pixel_matrix`: ``stack`` -- the FOV's channel stack when the caller has
    already read it; ``post_rownorm_q`` -- return ``(table, per-channel non-zero quantile Series)``, the
    quantile taken on the device; ``writer`` -- a ``TableWriter`` that takes over the two writes."""
    rows = _fov_device_half(tiff_dir, seg_dir, seg_suffix, img_sub_folder, is_mibitiff, channels, blur_factor,
                            pixel_thresh_val, channel_norm_df, fov, stack=stack, post_rownorm_q=post_rownorm_q)
    full, series = _fov_table_half(base_dir, data_dir, subset_dir, channels, subset_proportion, seed, fov, rows,
                                   writer=writer)
    return full if post_rownorm_q is None else (full, series)


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

    # Under a process group (torchrun, one process per GPU) the FOVs are dealt out by rank: every rank makes the
    # tables of its share, the per-FOV values behind the three normalisation files are gathered and averaged in one
    # agreed order, rank 0 writes those files.  Rank 0 alone looks at what is on disk; the others follow its plan.
    rank, world = distributed.init_from_env()
    ranks = (rank, world)
    data_root, subset_root = os.path.join(base_dir, data_dir), os.path.join(base_dir, subset_dir)
    pre_norm_file = os.path.join(out_root, norm_vals_name_pre_rownorm)
    thresh_file = os.path.join(out_root, pixel_thresh_name)
    per_fov_file = os.path.join(data_root, _PER_FOV_QUANTILES)

    plan = None
    if rank == 0:
        for folder in (data_root, subset_root):
            os.makedirs(folder, exist_ok=True)
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
            plan = {"todo": []}
        else:
            per_fov = pd.read_csv(per_fov_file, index_col="channel") if os.path.exists(per_fov_file) else pd.DataFrame()
            for name in sorted(os.listdir(data_root)):        # records the ranks of an interrupted sharded run left
                if name.startswith(_PER_FOV_QUANTILES + ".rank"):
                    part = pd.read_csv(os.path.join(data_root, name), index_col="channel")
                    fresh = [col for col in part.columns if col not in per_fov.columns]
                    if fresh:
                        per_fov = part[fresh] if per_fov.empty else per_fov.merge(part[fresh], how="outer",
                                                                                  left_index=True, right_index=True)
            todo = list(todo | (set(fovs) - set(per_fov.columns)))
            if len(todo) < len(fovs):
                print("Restarting preprocessing from FOV %s, "
                      "%d fovs left to process" % (todo[0], len(todo)))
            pixel_cluster_utils.check_for_modified_channels(tiff_dir=tiff_dir, test_fov=fovs[0],
                                                            img_sub_folder=img_sub_folder, channels=channels)
            plan = {"todo": todo, "per_fov": per_fov, "pre_norm": os.path.exists(pre_norm_file),
                    "thresh": os.path.exists(thresh_file)}
    plan = distributed.broadcast_object(plan, 0)
    if not plan["todo"]:
        return
    cohort_todo, per_fov = plan["todo"], plan["per_fov"]

    # up to three passes read the same TIFFs (two percentile passes, then the tables): decoded stacks are kept
    # on the host between them while they fit the cache budget, and each pass reads one FOV ahead
    stacks = image_io.stack_cache()
    if plan["pre_norm"]:
        pre_norm = read_dataframe(pre_norm_file)
    else:
        pre_norm = pixel_cluster_utils.calculate_channel_percentiles(
            tiff_dir=tiff_dir, fovs=fovs, channels=channels, img_sub_folder=img_sub_folder,
            percentile=channel_percentile_pre_rownorm, stacks=stacks, ranks=ranks)
        if rank == 0:
            write_dataframe(pre_norm, pre_norm_file, compression='uncompressed')

    if plan["thresh"]:
        pixel_thresh_val = read_dataframe(thresh_file)['pixel_thresh_val'].values[0]
    else:
        pixel_thresh_val = pixel_cluster_utils.calculate_pixel_intensity_percentile(
            tiff_dir=tiff_dir, fovs=fovs, channels=channels, img_sub_folder=img_sub_folder,
            channel_percentiles=pre_norm, stacks=stacks, ranks=ranks)
        if rank == 0:
            write_dataframe(pd.DataFrame({'pixel_thresh_val': [pixel_thresh_val]}), thresh_file,
                            compression='uncompressed')

    # this rank's share of the tables; with several ranks each keeps the restart record of its own FOVs
    todo = cohort_todo if world == 1 else distributed.shard(cohort_todo, rank, world)
    if world > 1:
        per_fov_file = per_fov_file + ".rank%d" % rank
        per_fov = pd.DataFrame()

    # Two stages per FOV, one FOV apart: the caller's thread drives the device (segmentation, channel division,
    # kernels, rows back) while a finisher thread turns the previous FOV's rows into the two DataFrames (seeded
    # sample included: the finisher is the only user of numpy's global RNG meanwhile) and hands them, and strictly
    # behind them the per-FOV record, to the table writers (four: serialising a FOV's tables takes longer than
    # either stage); the record never reaches the disk before the tables it vouches for.
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor
    import threading
    writer = TableWriter(depth=8, workers=4)
    record = {"per_fov": per_fov, "snapshots": {}, "written": set(), "next": 0, "count": 0}
    record_lock = threading.Lock()

    def tables_written(index):
        """Writer threads, once per FOV when both of its tables are out: the per-FOV record goes to disk for the
        longest run of finished FOVs (in processing order), never ahead of a table it vouches for."""
        if writer.failed:           # the callback also runs for tables abandoned after a failed write
            return
        with record_lock:
            record["written"].add(index)
            newest = None
            while record["next"] in record["written"]:
                newest = record["snapshots"].pop(record["next"])
                record["next"] += 1
            if newest is not None:
                newest.to_csv(per_fov_file)

    def finish(fov, rows):
        index = record["count"]
        record["count"] += 1
        left = [2]

        def one_written():
            with record_lock:
                left[0] -= 1
                last = left[0] == 0
            if last:
                tables_written(index)

        values, kept_h, width, quantiles, labels = rows[:5]
        release = rows[5] if len(rows) > 5 else None
        full, subset = _assemble_tables(fov, channels, values, kept_h, width, labels, subset_proportion, seed=seed)
        row = pd.Series(quantiles, index=pd.Index(list(channels), name="channel"), name=fov)
        table = record["per_fov"]
        if fov in table.columns:                  # re-done after an interrupted run
            table = table.drop(columns=[fov])
        record["per_fov"] = table = table.merge(row, how="outer", left_index=True, right_index=True)
        with record_lock:
            record["snapshots"][index] = table.copy()      # after every FOV: an interrupted run keeps what it has
        def full_written():            # the full table wraps the recycled block; the sub-sample owns its rows
            if release is not None:
                release()
            one_written()

        writer.submit(subset, os.path.join(base_dir, subset_dir, fov + ".feather"), done=one_written)
        writer.submit(full, os.path.join(base_dir, data_dir, fov + ".feather"), done=full_written)

    group = batch_size if multiprocess else 1
    done = 0
    ahead = image_io.iter_stacks(tiff_dir, todo, channels, img_sub_folder, cache=stacks, fill=False)
    finisher = ThreadPoolExecutor(max_workers=1, thread_name_prefix="pxsom-fov-tables")
    staging = flowsom.HostStaging()
    blocks = HostBlocks()
    pending = deque()
    try:
        for start in range(0, len(todo), group):
            names = todo[start:start + group]
            for fov in names:
                stack = next(ahead)[1]
                stacks.pop(fov, None)     # last use of this FOV's stack
                rows = _fov_device_half(tiff_dir, seg_dir, seg_suffix, img_sub_folder, is_mibitiff, channels,
                                        blur_factor, pixel_thresh_val, pre_norm, fov, stack=stack,
                                        post_rownorm_q=channel_percentile_post_rownorm, staging=staging,
                                        blocks=blocks)
                while len(pending) >= 2:              # at most two FOVs' rows wait for the finisher
                    pending.popleft().result()
                pending.append(finisher.submit(finish, fov, rows))
            done += len(names)
            while (multiprocess or done == len(todo)) and pending:   # a batch / the cohort is reported once its
                pending.popleft().result()                           # tables are handed to the writer
            if world == 1 and (multiprocess or done % 10 == 0 or done == len(todo)):
                print("Processed %d fovs" % done)
    finally:
        finisher.shutdown(wait=True)
        writer.close()
    per_fov = record["per_fov"]
    if world > 1:
        # the ranks' new per-FOV values join what was on record, in the order of the agreed to-do list
        parts = distributed.allgather_objects(per_fov)
        if rank != 0:
            distributed.barrier()
            return
        per_fov = plan["per_fov"]
        for fov in cohort_todo:
            for part in parts:
                if fov in part.columns:
                    if fov in per_fov.columns:            # re-done after an interrupted run
                        per_fov = per_fov.drop(columns=[fov])
                    per_fov = part[[fov]] if per_fov.empty else per_fov.merge(part[[fov]], how="outer",
                                                                            left_index=True, right_index=True)
        print("Processed %d fovs" % len(cohort_todo))

    # cohort value per channel = mean of the per-FOV values; channels in natural order
    cohort = pd.DataFrame(per_fov.mean(axis=1))
    cohort = cohort.loc[sorted(cohort.index, key=natsort_key)]
    write_dataframe(cohort.T, os.path.join(base_dir, norm_vals_name_post_rownorm), compression='uncompressed')
    for name in os.listdir(data_root):                     # the record (and the ranks' own) has done its job
        if name.startswith(_PER_FOV_QUANTILES):
            os.remove(os.path.join(data_root, name))
    if world > 1:
        distributed.barrier()
