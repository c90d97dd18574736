"""Shim for alpineer.io_utils (test-infra only)."""
from ark_analysis_amd.host_utils import (list_files, list_folders, remove_file_extensions,  # noqa
                                          validate_paths)
