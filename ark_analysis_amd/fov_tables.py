"""Per-FOV feather tables on disk -- the wire format between the Pixie stages -- and the machinery the
pipeline functions share for walking them.

Reference contract (read for behaviour, not structure): one ``<fov>.feather`` (Arrow IPC, written
uncompressed) per field of view inside ``<base_dir>/<data_dir>``; a stage that rewrites the tables
writes into ``<data_dir>_temp`` and swaps the directories when every FOV is done, so an interrupted
run restarts from the FOVs that are still missing in ``_temp``
(/root/reference/src/ark/phenotyping/pixel_som_clustering.py:114-134, :288-289;
pixel_cluster_utils.py:419-478).

What is different here, on purpose (SURVEY.md section 8 f, rank 1): with the BMU search at HBM speed the
stage is I/O bound, so

* the *next* tables are read by a background thread while the current one is on the GPU
  (:class:`TablePrefetcher`), and finished tables are written by another thread (:class:`TableWriter`);
* "which FOVs still lack column X" looks at the Arrow schema in the file footer instead of loading a
  whole table (:meth:`FovTableDir.column_names`).
"""
import collections
import os
import queue
import shutil
import tempfile
import threading
import warnings
from concurrent.futures import ThreadPoolExecutor
from typing import Iterable, Iterator, List, Optional, Sequence, Tuple

import pandas as pd
import pyarrow as pa
from pyarrow.lib import ArrowInvalid

from .host_utils import natsorted

#: what a damaged table raises when opened (the reference catches exactly these)
UNREADABLE = (ArrowInvalid, OSError, IOError)

SUFFIX = ".feather"
#: bookkeeping columns that are never SOM features
POSITION_COLUMNS = ("fov", "row_index", "column_index")
OPTIONAL_COLUMNS = ("label", "pixel_som_cluster", "pixel_meta_cluster", "pixel_meta_cluster_rename")


def read_table(path) -> pa.Table:
    """The file's Arrow table (column chunks as stored; nothing is converted)."""
    with pa.OSFile(str(path), "rb") as src:
        return pa.ipc.open_file(src).read_all()


def read_dataframe(path) -> pd.DataFrame:
    """What ``feather.read_dataframe`` does: a Feather V2 file *is* an Arrow IPC file."""
    return read_table(path).to_pandas()


def write_dataframe(df, path, compression="uncompressed") -> None:
    """What ``feather.write_dataframe(df, path, compression=...)`` does (64 Ki-row record batches).
    ``df`` may also be an Arrow table that already carries pandas metadata."""
    body = df if isinstance(df, pa.Table) else pa.Table.from_pandas(df, preserve_index=None)
    options = pa.ipc.IpcWriteOptions(compression=None if compression in (None, "uncompressed") else compression)
    with pa.OSFile(str(path), "wb") as sink, pa.ipc.new_file(sink, body.schema, options=options) as out:
        out.write_table(body, max_chunksize=1 << 16)


def unify_label_column(table: pd.DataFrame) -> pd.DataFrame:
    """Tables written by newer pipeline versions call the segmentation id ``segmentation_label``."""
    if "segmentation_label" in table.columns:
        table.rename(columns={"segmentation_label": "label"}, inplace=True)
    return table


def feature_columns(table: pd.DataFrame) -> pd.Index:
    """Columns left once position, label and cluster columns are removed (order preserved)."""
    skip = set(POSITION_COLUMNS) | set(OPTIONAL_COLUMNS)
    return table.columns[[c not in skip for c in table.columns]]


class FovTableDir:
    """A directory of ``<fov>.feather`` tables plus its ``_temp`` staging twin."""

    def __init__(self, root):
        self.root = str(root)
        self.staging = self.root + "_temp"

    # ---- naming -----------------------------------------------------------------------------
    @staticmethod
    def _tables_in(folder) -> List[str]:
        found = [name for name in os.listdir(folder)
                 if SUFFIX in name and not name.startswith(".")
                 and not os.path.isdir(os.path.join(folder, name))]
        return natsorted(found)

    def files(self) -> List[str]:
        return self._tables_in(self.root)

    def fovs(self) -> List[str]:
        return [os.path.splitext(name)[0] for name in self.files()]

    def path(self, fov: str, staged: bool = False) -> str:
        return os.path.join(self.staging if staged else self.root, fov + SUFFIX)

    # ---- reading ----------------------------------------------------------------------------
    def load(self, fov: str) -> pd.DataFrame:
        return read_dataframe(self.path(fov))

    def load_arrow(self, fov: str) -> pa.Table:
        return read_table(self.path(fov))

    def column_names(self, fov: str) -> List[str]:
        """Column names from the Arrow footer only (no data pages are touched)."""
        with pa.OSFile(self.path(fov), "rb") as src:
            return list(pa.ipc.open_file(src).schema.names)

    def first_readable(self) -> Optional[pd.DataFrame]:
        """The first table that opens; damaged files are passed over.  ``None`` if none opens."""
        for name in self.files():
            try:
                return read_dataframe(os.path.join(self.root, name))
            except UNREADABLE:
                continue
        return None

    # ---- restartable rewrite ----------------------------------------------------------------
    def pending(self, column: str) -> List[str]:
        """FOVs a stage that adds ``column`` still has to process.

        No staging directory yet: every FOV if the first readable table lacks ``column`` (the staging
        directory is created as a side effect, like the reference's helper does), nobody otherwise.
        Staging directory present (an earlier run was interrupted): the FOVs not yet staged."""
        if os.path.exists(self.staging):
            done = set(self._tables_in(self.staging))
            return [os.path.splitext(name)[0] for name in self.files() if name not in done]
        probe = self.first_readable()
        if probe is None:
            raise FileNotFoundError("no readable FOV table (*%s) in %s" % (SUFFIX, self.root))
        if column in probe.columns.values:
            return []
        os.mkdir(self.staging)
        return self.fovs()

    def open_staging(self) -> None:
        os.mkdir(self.staging)

    def commit(self, on_rm_error=None) -> None:
        """Replace the directory by its staging twin.  The old tables are moved aside (a hidden, uniquely named
        directory next to ``root``) and unlinked on a background thread the interpreter waits for at exit:
        deleting a page-cache-resident 218 MB table takes ~16 ms, a third of what labelling it costs, and nothing
        downstream depends on it -- ``root`` holds the new tables when this returns.  The public pipeline functions
        call :func:`wait_for_cleanup` before they return (a caller that deletes or lists the parent directory right
        after them must not race the cleaner); the deletion then overlaps only with what they do after the swap."""
        parent, name = os.path.split(os.path.abspath(self.root))
        # what an earlier process that was killed between its rename and the end of its clean-up left behind (a full
        # copy of the cohort's tables): swept here, behind the directories this process is still deleting itself
        wait_for_cleanup()
        for stale in os.listdir(parent):
            if stale.startswith(".%s.old-" % name) and os.path.isdir(os.path.join(parent, stale)):
                _remove_tree(os.path.join(parent, stale), on_rm_error)
        trash = tempfile.mkdtemp(prefix=".%s.old-" % name, dir=parent)
        os.rename(self.root, os.path.join(trash, name))
        shutil.move(self.staging, self.root)
        cleaner = threading.Thread(target=_remove_tree, args=(trash, on_rm_error), name="fov-cleanup", daemon=False)
        cleaner.start()
        _CLEANERS.append(cleaner)


_CLEANERS: List[threading.Thread] = []


def _remove_tree(path: str, on_rm_error) -> None:
    """Deletes a directory of replaced tables.  The files are unlinked side by side first (releasing a 218 MB page-cache
    resident table takes tens of milliseconds inside the kernel, and a cohort has hundreds of them), the rest -- empty
    directories, anything unexpected -- goes through shutil.rmtree with the caller's error hook."""
    try:
        files = [os.path.join(folder, name) for folder, _, names in os.walk(path) for name in names]
        if len(files) > 1:
            def unlink(file_path):
                try:
                    os.unlink(file_path)
                except OSError:
                    pass                      # left for rmtree below, which reports through on_rm_error
            with ThreadPoolExecutor(max_workers=min(16, len(files)), thread_name_prefix="fov-unlink") as pool:
                list(pool.map(unlink, files))
        shutil.rmtree(path, onerror=on_rm_error)
    except BaseException as err:      # nobody to raise to on this thread
        warnings.warn("could not remove the replaced tables in %s: %r" % (path, err))


def wait_for_cleanup() -> None:
    """Blocks until every directory replaced by :meth:`FovTableDir.commit` so far has been deleted."""
    while _CLEANERS:
        _CLEANERS.pop().join()


class TablePrefetcher:
    """Iterates ``(fov, table-or-None)`` over ``fovs`` in order, with up to ``depth`` tables being read ahead on
    ``workers`` background threads (as DataFrames, or as Arrow tables with ``as_arrow``).  ``None`` stands for
    a table that could not be opened."""

    def __init__(self, tables: FovTableDir, fovs: Sequence[str], depth: int = 2, as_arrow: bool = False,
                 workers: int = 1):
        self._fovs = list(fovs)
        self._read = tables.load_arrow if as_arrow else tables.load
        self._depth = max(1, depth)
        self._stop = threading.Event()
        self._pool = ThreadPoolExecutor(max_workers=max(1, workers), thread_name_prefix="fov-prefetch")
        self._ahead: "collections.deque" = collections.deque()    # (fov, future), in FOV order
        self._next = 0
        self._top_up()

    def _load(self, fov):
        if self._stop.is_set():
            return None
        try:
            return self._read(fov)
        except UNREADABLE:
            return None

    def _top_up(self) -> None:
        while len(self._ahead) < self._depth and self._next < len(self._fovs) and not self._stop.is_set():
            fov = self._fovs[self._next]
            self._next += 1
            self._ahead.append((fov, self._pool.submit(self._load, fov)))

    def __iter__(self) -> Iterator[Tuple[str, Optional[pd.DataFrame]]]:
        while self._ahead:
            fov, pending = self._ahead.popleft()
            self._top_up()
            yield fov, pending.result()       # a reader's exception (other than "unreadable") surfaces here

    def close(self) -> None:
        """Stop reading ahead and drop what is queued (call from a ``finally``: a consumer that leaves early --
        a GPU error, an exception in the caller -- must not leave readers holding tables)."""
        self._stop.set()
        for _, pending in self._ahead:
            pending.cancel()
        self._ahead.clear()
        self._pool.shutdown(wait=True)


class TableWriter:
    """Writes tables on background threads; ``close()`` waits and re-raises the first failure.  One worker
    (the default) writes in submission order -- which :meth:`submit_call` relies on; several workers write
    independent tables side by side (a 218 MB table takes ~60 ms to serialise: one writer caps a stage that
    labels a FOV in 25 ms)."""

    def __init__(self, depth: int = 2, workers: int = 1):
        self._jobs: "queue.Queue" = queue.Queue(maxsize=max(1, depth))
        self._error: Optional[BaseException] = None
        self._ordered = workers <= 1
        self._workers = [threading.Thread(target=self._drain, name="fov-write-%d" % i, daemon=True)
                         for i in range(max(1, workers))]
        for worker in self._workers:
            worker.start()

    def _drain(self) -> None:
        while True:
            job = self._jobs.get()
            if job is None:
                return
            if self._error is None:
                try:
                    if callable(job):
                        job()
                    else:
                        write_dataframe(job[0], job[1], compression="uncompressed")
                except BaseException as err:
                    self._error = err
            if not callable(job) and job[2] is not None:
                job[2]()     # the table's buffers may be reused from here on (written, or abandoned after a failure)

    @property
    def failed(self) -> bool:
        """True once a write has failed (``done`` callbacks still run for the jobs abandoned after it)."""
        return self._error is not None

    def _raise_if_failed(self) -> None:
        if self._error is not None:   # stop the stage at the first failed write, not at close()
            raise self._error

    def submit(self, table, path: str, done=None) -> None:
        """``done()`` is called on the writer thread once the table is no longer needed."""
        self._raise_if_failed()
        self._jobs.put((table, path, done))

    def submit_call(self, fn) -> None:
        """Run ``fn()`` on the writer thread after everything submitted before it (e.g. a progress record
        that must not reach the disk before the tables it describes).  Single-worker writers only."""
        if not self._ordered:
            raise RuntimeError("submit_call needs a single-worker TableWriter (ordered writes)")
        self._raise_if_failed()
        self._jobs.put(fn)

    def close(self) -> None:
        for _ in self._workers:
            self._jobs.put(None)
        for worker in self._workers:
            worker.join()
        if self._error is not None:
            raise self._error


def batches(items: Sequence, size: int) -> Iterable[Sequence]:
    for start in range(0, len(items), size):
        yield items[start:start + size]
