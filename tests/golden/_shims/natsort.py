"""Shim for natsort (test-infra only): natural-sort key / sorted."""
from ark_analysis_amd.host_utils import natsort_key, natsorted  # noqa: F401
