"""ark_analysis_amd -- MI355X (gfx950) implementation of ark-analysis' Pixie pixel/cell SOM hot path.

Layout
  csrc/ + include/pxsom.h   hand-written HIP kernels behind a C ABI (libpxsom.so)
  _capi, som_device          ctypes binding / torch-tensor plumbing
  flowsom                    pyFlowSOM-compatible ``som`` / ``map_data_to_nodes`` (the two foreign
                             calls the reference makes, cluster_helpers.py:106-109, 152-157)
  distributed                FOV-sharded batch SOM training with per-step RCCL all-reduce
  phenotyping/               mirrors of the reference's pipeline modules (same names/signatures)

(The directory is ``ark_analysis_amd`` rather than ``ark-analysis_amd`` because a Python package
name cannot contain a hyphen.)
"""
__version__ = "0.1.0"
