"""Import stand-in (test-infra only) for the names ark.utils.data_utils imports at module level."""


class AnnData:
    pass


def read_zarr(*a, **k):
    raise NotImplementedError("anndata is not in this image")
