"""ctypes binding of libpxsom.so (the C ABI declared in include/pxsom.h).

Device memory, streams and process groups come from PyTorch-ROCm; the arithmetic is in the
hand-written HIP kernels behind this boundary.  There is no CPU fallback: if the library is
missing it is built with hipcc, and if that fails the import raises.
"""
import ctypes
import os

import torch  # noqa: F401  (loads torch's libamdhip64 first so libpxsom binds to the same runtime)

from . import _build

PXSOM_F32 = 0
PXSOM_F64 = 1
PXSOM_F16 = 2
MAX_CHANNELS = 1024
MAX_NODES = 1024

_STATUS = {0: "PXSOM_OK", -1: "PXSOM_ERR_INVALID_ARG", -2: "PXSOM_ERR_UNSUPPORTED",
           -3: "PXSOM_ERR_WORKSPACE", -4: "PXSOM_ERR_HIP"}

# every symbol include/pxsom.h declares: (restype, argtypes)
_vp, _i32, _i64, _f64, _sz = (ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_double,
                              ctypes.c_size_t)
ABI_VERSION = 9  # include/pxsom.h PXSOM_ABI_VERSION

SYMBOLS = {
    "pxsom_abi_version": (_i32, []),
    "pxsom_last_error": (ctypes.c_char_p, []),
    "pxsom_prof_create": (_i32, [ctypes.POINTER(ctypes.c_void_p)]),
    "pxsom_prof_destroy": (_i32, [_vp]),
    "pxsom_prof_attach": (_i32, [_vp, _i64]),
    "pxsom_prof_collect": (_i32, [_vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64)]),
    "pxsom_host_glibc_rand_fill": (_i32, [ctypes.c_uint32, _i64, _vp]),
    "pxsom_assign_workspace_bytes": (_sz, [_i64, _i32, _i32]),
    "pxsom_assign": (_i32, [_vp, _i64, _i32, _i64, _i32, _vp, _i32, _vp, _vp, _vp, _sz, _vp]),
    "pxsom_assign_ex": (_i32, [_vp, _i64, _i32, _i64, _i32, _vp, _i32, _vp, _vp, _vp, _sz, _i32, _vp]),
    "pxsom_assign_last_exact_rows": (_i32, [_vp, _vp, ctypes.POINTER(ctypes.c_int64)]),
    "pxsom_cluster_sums": (_i32, [_vp, _i64, _i32, _i64, _i32, _vp, _i32, _vp, _vp, _vp]),
    "pxsom_train_online": (_i32, [_vp, _i64, _i32, _i64, _i32, _vp, _i32, _i32, _i32, _f64, _f64,
                                  _f64, _f64, _vp, _vp]),
    "pxsom_train_online_ex": (_i32, [_vp, _i64, _i32, _i64, _i32, _vp, _i32, _i32, _i32, _f64, _f64,
                                     _f64, _f64, _vp, _i32, _vp]),
    "pxsom_batch_accumulate": (_i32, [_vp, _i64, _i32, _i64, _i32, _vp, _i32, _vp, _vp, _vp, _sz, _i32, _vp]),
    "pxsom_quantile_f32": (_i32, [_vp, _i64, _i32, _i64, _f64, _i32, _vp, _vp, _sz, _vp]),
    "pxsom_scaled_rowsum_f32": (_i32, [_vp, _i64, _i32, _i64, _vp, _vp, _vp]),
    "pxsom_scaled_rowsum_f64": (_i32, [_vp, _i64, _i32, _i64, _vp, _vp, _vp]),
    "pxsom_cluster_mask_workspace_bytes": (_sz, [_i32, _i32]),
    "pxsom_cluster_mask": (_i32, [_vp, _vp, _vp, _i64, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _sz, _vp]),
    "pxsom_pair_histogram": (_i32, [_vp, _vp, _i64, _i64, _i32, _vp, _vp]),
    "pxsom_batch_update_prepare": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _f64, _f64, _vp, _sz, _vp]),
    "pxsom_gaussian_blur_hwc": (_i32, [_vp, _vp, _i32, _i32, _i32, _vp, _i32, _i32, _vp]),
    "pxsom_rownorm_workspace_bytes": (_sz, [_i64]),
    "pxsom_rowsum_filter_normalize": (_i32, [_vp, _i64, _i32, _f64, _vp, _vp, _vp, _vp, _sz, _i32, _vp]),
    "pxsom_normalize_columns": (_i32, [_vp, _i64, _i32, _i64, _vp, _vp, _i64, _vp]),
    "pxsom_quantile_workspace_bytes": (_sz, [_i64, _i32]),
    "pxsom_quantile_nonzero": (_i32, [_vp, _i64, _i32, _i64, _f64, _i32, _vp, _vp, _sz, _vp]),
    "pxsom_batch_update": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _f64, _f64, _vp]),
    "pxsom_assign_sums_workspace_bytes": (_sz, [_i64, _i32, _i32]),
    "pxsom_assign_sums": (_i32, [_vp, _i64, _i32, _i64, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "pxsom_assign_means": (_i32, [_vp, _i64, _i32, _i64, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "pxsom_assign_sums_scratch_bytes": (_sz, [_i32, _i32]),
    "pxsom_assign_sums_ex": (_i32, [_vp, _i64, _i32, _i64, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _sz, _i32, _vp]),
    "pxsom_assign_means_ex": (_i32, [_vp, _i64, _i32, _i64, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _sz, _i32, _vp]),
    "pxsom_relabel": (_i32, [_vp, _i64, _vp, _i32, _i32, _vp, _vp]),
    "pxsom_batch_train_workspace_bytes": (_sz, [_i64, _i32, _i32, _i32]),
    "pxsom_batch_train_steps": (_i32, [_vp, _i64, _i32, _i64, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32,
                                       _f64, _f64, _f64, _f64, _vp, _sz, _i32, _vp]),
    "pxsom_batch_train_finish": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _f64, _f64, _f64, _f64, _vp, _vp]),
    "pxsom_batch_train_steps_sharded": (_i32, [_vp, _i64, _i32, _i64, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32,
                                               _f64, _f64, _f64, _f64, _vp, _sz, _i32, _vp, _vp]),
    "pxsom_batch_train_sched_workspace_bytes": (_sz, [_i64, _i32, _i32, _i32, _i32, _vp, _i32]),
    "pxsom_batch_train_sched": (_i32, [_vp, _i64, _i32, _i64, _i32, _vp, _vp, _i32, _i32, _i32, _vp, _i32, _i32, _i32, _i32,
                                       _f64, _f64, _f64, _f64, _f64, _vp, _sz, _i32, _vp, _vp]),
    "pxsom_batch_train_sched_from": (_i32, [_vp, _i64, _i32, _i64, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _i32, _i32, _i32, _i32,
                                            _f64, _f64, _f64, _f64, _f64, _vp, _sz, _i32, _vp, _vp]),
    "pxsom_exact_sum_quantum": (_f64, [_f64, _i64]),
    "pxsom_absmax": (_i32, [_vp, _i64, _i32, _i64, _i32, _vp, _vp]),
    "pxsom_batch_train_sched_finish": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _i32, _f64, _f64, _f64, _f64,
                                              _vp, _vp]),
    "pxsom_batch_train_fused_route": (_i32, [_vp, _i32, _i64, _i32, _i32, _i32, _i32]),
    "pxsom_comm_bind": (_i32, [ctypes.c_char_p]),
    "pxsom_comm_unique_id": (_i32, [_vp, _sz]),
    "pxsom_comm_create": (_i32, [_vp, _sz, _i32, _i32, ctypes.POINTER(_vp)]),
    "pxsom_comm_destroy": (_i32, [_vp]),
    "pxsom_comm_allreduce_sum_f64": (_i32, [_vp, _vp, _sz, _vp]),
    "pxsom_comm_p2p_create": (_i32, [_i32, _i32, _sz, ctypes.POINTER(_vp)]),
    "pxsom_comm_p2p_handle": (_i32, [_vp, _vp, _sz]),
    "pxsom_comm_p2p_connect": (_i32, [_vp, _vp, _sz]),
    "pxsom_comm_p2p_error": (_i32, [_vp, ctypes.POINTER(ctypes.c_uint64)]),
    "pxsom_comm_p2p_set_fused": (_i32, [_vp, _i32]),
}

_lib = None


class PxsomError(RuntimeError):
    """A libpxsom call returned a non-zero status."""


def library_path() -> str:
    return _build.SO_PATH


def lib():
    """Load (building if needed) libpxsom.so and declare every prototype."""
    global _lib
    if _lib is None:
        path = _build.build() if _build.needs_build() else _build.SO_PATH
        if not os.path.exists(path):
            raise ImportError(f"libpxsom.so not found at {path} and could not be built")
        L = ctypes.CDLL(path)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        if L.pxsom_abi_version() != ABI_VERSION:
            raise ImportError(f"libpxsom.so ABI {L.pxsom_abi_version()} != {ABI_VERSION}")
        _lib = L
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().pxsom_last_error()
        raise PxsomError(f"{what}: {_STATUS.get(rc, rc)}: {msg.decode(errors='replace') if msg else ''}")


def require_gpu() -> torch.device:
    """The product path is HIP-only: fail loudly when there is no GPU."""
    if not torch.cuda.is_available():
        raise RuntimeError(
            "ark_analysis_amd: no HIP device visible -- the SOM kernels only exist as gfx950 HIP "
            "code (libpxsom.so); there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return PXSOM_F32
    if t.dtype == torch.float64:
        return PXSOM_F64
    if t.dtype == torch.float16:
        return PXSOM_F16
    raise TypeError(f"pixel matrix must be float16, float32 or float64, got {t.dtype}")


def glibc_rand(seed: int, count: int):
    import numpy as np
    out = np.empty(int(count), dtype=np.int32)
    check(lib().pxsom_host_glibc_rand_fill(int(seed) & 0xFFFFFFFF, int(count),
                                           out.ctypes.data), "pxsom_host_glibc_rand_fill")
    return out


class KernelTimer:
    """HIP-event timer around the BMU filter kernel of every assign call with >= min_rows rows."""

    def __init__(self, min_rows: int = 0):
        self._h = ctypes.c_void_p()
        check(lib().pxsom_prof_create(ctypes.byref(self._h)), "pxsom_prof_create")
        self.min_rows = int(min_rows)

    def __enter__(self):
        check(lib().pxsom_prof_attach(self._h, self.min_rows), "pxsom_prof_attach")
        return self

    def __exit__(self, *exc):
        lib().pxsom_prof_attach(None, 0)

    def collect(self):
        """(total_ms, launches) since the last collect/attach; synchronises."""
        ms, cnt = ctypes.c_double(0), ctypes.c_int64(0)
        check(lib().pxsom_prof_collect(self._h, ctypes.byref(ms), ctypes.byref(cnt)),
              "pxsom_prof_collect")
        return ms.value, cnt.value

    def __del__(self):
        try:
            lib().pxsom_prof_destroy(self._h)
        except Exception:
            pass
