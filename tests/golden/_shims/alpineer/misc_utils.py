"""Shim for alpineer.misc_utils (test-infra only)."""
from ark_analysis_amd.host_utils import verify_in_list, verify_same_elements  # noqa: F401
