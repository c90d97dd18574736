"""Shim for natsort (test-infra only): natural-sort key / sorted.  Like the real ``natsort_key``: a string
maps to a flat tuple of its text and integer runs (``'chan10' -> ('chan', 10)``), a non-string iterable to
the tuple of its elements' keys (pandas hands ``sort_index(key=...)`` a whole Index)."""
import re

from ark_analysis_amd.host_utils import natsorted  # noqa: F401

_RUNS = re.compile(r"(\d+)")


def natsort_key(value):
    if isinstance(value, (str, bytes)) or not hasattr(value, "__iter__"):
        parts = _RUNS.split(str(value))
        flat = [int(p) if i % 2 else p for i, p in enumerate(parts)]
        if flat and flat[-1] == "":
            flat.pop()
        return tuple(flat)
    return tuple(natsort_key(v) for v in value)
