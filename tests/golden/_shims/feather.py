"""Shim: feather-format 0.4.x is a re-export of pyarrow.feather (test-infra only)."""
from pyarrow.feather import read_feather as read_dataframe


def write_dataframe(df, dest, compression=None, **kw):
    from pyarrow.feather import write_feather
    write_feather(df, dest, compression=compression, **kw)
