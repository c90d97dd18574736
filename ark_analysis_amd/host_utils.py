"""Host-side path / list / natural-sort helpers.

The reference leans on two small third-party packages for these (``alpineer.io_utils``,
``alpineer.misc_utils``, ``natsort``; imported at e.g.
``/root/reference/src/ark/phenotyping/pixel_som_clustering.py:8`` and
``cluster_helpers.py:8-13``).  Neither is installed in this image, so the handful of
functions the hot path touches are written here from their documented behaviour:
same names, same argument meaning, same exception types, so that the mirrored
pipeline functions and their tests read like the reference's.

One deliberate difference: :func:`list_files` returns a *naturally sorted* list
(``os.listdir`` order is filesystem dependent, and the reference builds the SOM training
matrix in that order, ``cluster_helpers.py:211-215``); sorting makes the pixel presentation
order -- an explicit input of the SOM oracle -- reproducible.
"""
import os
import pathlib
import re
from typing import Iterable, List, Optional, Sequence, Union

_DIGITS = re.compile(r"(\d+)")


def natsort_key(s) -> tuple:
    """Natural-sort key: digit runs compare as integers (``chan2`` < ``chan10``)."""
    parts = _DIGITS.split(str(s))
    return tuple((1, int(p), "") if i % 2 else (0, 0, p) for i, p in enumerate(parts))


def natsorted(seq: Iterable) -> list:
    return sorted(seq, key=natsort_key)


def validate_paths(paths: Union[str, os.PathLike, Sequence]) -> None:
    """Raise ``FileNotFoundError`` if any path does not exist (alpineer.io_utils.validate_paths)."""
    if isinstance(paths, (str, os.PathLike)):
        paths = [paths]
    for path in paths:
        if not os.path.exists(path):
            p = pathlib.Path(path)
            for parent in reversed(p.parents):
                if not os.path.exists(parent):
                    raise FileNotFoundError(
                        f"A bad path, {path}, was provided.\n"
                        f"The folder, {parent.name}, could not be found...")
            raise FileNotFoundError(
                f"The file/path, {p.name}, could not be found in {p.parent}")


def list_files(dir_name, substrs: Optional[Union[str, List[str]]] = None,
               exact_match: bool = False, ignore_hidden: bool = True) -> List[str]:
    """Files (not directories) in ``dir_name`` whose name contains / equals any of ``substrs``."""
    files = [f for f in os.listdir(dir_name) if not os.path.isdir(os.path.join(dir_name, f))]
    if ignore_hidden:
        files = [f for f in files if not f.startswith(".")]
    files = natsorted(files)
    if substrs is None:
        return files
    if not isinstance(substrs, (list, tuple)):
        substrs = [substrs]
    if exact_match:
        return [f for f in files if any(s == os.path.splitext(f)[0] for s in substrs)]
    return [f for f in files if any(s in f for s in substrs)]


def list_folders(dir_name, substrs=None, exact_match: bool = False,
                 ignore_hidden: bool = True) -> List[str]:
    folders = [f for f in os.listdir(dir_name) if os.path.isdir(os.path.join(dir_name, f))]
    if ignore_hidden:
        folders = [f for f in folders if not f.startswith(".")]
    folders = natsorted(folders)
    if substrs is None:
        return folders
    if not isinstance(substrs, (list, tuple)):
        substrs = [substrs]
    if exact_match:
        return [f for f in folders if any(s == f for s in substrs)]
    return [f for f in folders if any(s in f for s in substrs)]


def remove_file_extensions(files: Optional[List[str]]) -> Optional[List[str]]:
    if files is None:
        return None
    return [os.path.splitext(f)[0] for f in files]


def _as_list(v) -> list:
    if v is None:
        return []
    if isinstance(v, (str, bytes)) or not hasattr(v, "__iter__"):
        return [v]
    return list(v)


def verify_in_list(warn: bool = False, **kwargs) -> bool:
    """``verify_in_list(a=xs, b=ys)``: every element of ``xs`` must be in ``ys`` (else ValueError)."""
    if len(kwargs) != 2:
        raise ValueError("You must provide 2 arguments to verify_in_list")
    (test_name, test_list), (good_name, good_values) = kwargs.items()
    test_list, good_values = _as_list(test_list), _as_list(good_values)
    good = set(good_values)
    bad = [v for v in test_list if v not in good]
    if bad:
        msg = ("Not all values given in list {0} were found in list {1}.\n "
               "Displaying {2} of {3} invalid value(s) for list {0}\n{4}").format(
                   test_name, good_name, min(len(bad), 10), len(bad), bad[:10])
        if warn:
            import warnings
            warnings.warn(msg)
            return False
        raise ValueError(msg)
    return True


def verify_same_elements(enforce_order: bool = False, warn: bool = False, **kwargs) -> bool:
    """Both lists must hold the same elements (and, with ``enforce_order``, in the same order)."""
    if len(kwargs) != 2:
        raise ValueError("You must provide 2 list arguments to verify_same_elements")
    (n1, l1), (n2, l2) = kwargs.items()
    try:
        l1, l2 = list(l1), list(l2)
    except TypeError:
        raise ValueError("Both arguments provided must be lists or list types")
    msg = None
    if set(l1) != set(l2):
        msg = (f"Lists {n1} and {n2} are not identical: "
               f"only in {n1}: {[v for v in l1 if v not in set(l2)][:10]}, "
               f"only in {n2}: {[v for v in l2 if v not in set(l1)][:10]}")
    elif enforce_order and l1 != l2:
        first = next(i for i, (a, b) in enumerate(zip(l1, l2)) if a != b)
        msg = (f"Lists {n1} and {n2} ordered differently: values {l1[first]} and {l2[first]} "
               f"do not match at index {first}")
    if msg:
        if warn:
            import warnings
            warnings.warn(msg)
            return False
        raise ValueError(msg)
    return True
