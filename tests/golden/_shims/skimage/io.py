"""Shim (unused by the hot path)."""


def imread(*a, **k):
    raise NotImplementedError("skimage shim")
