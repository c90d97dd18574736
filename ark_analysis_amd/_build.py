"""Builds libpxsom.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

The shared object sits next to this file so that it travels with the source tree (gpurun
snapshots, no site-packages install).  hipcc cross-compiles without a GPU.
"""
import fcntl
import hashlib
import os
import shutil
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
SO_PATH = os.path.join(_PKG, "libpxsom.so")
SOURCES = ["pxsom_api.hip", "pxsom_assign.hip", "pxsom_assign_filter.hip", "pxsom_assign_filter_acc.hip", "pxsom_batch_step.hip", "pxsom_batch_step_wide.hip", "pxsom_train.hip",
           "pxsom_pre.hip", "pxsom_sums.hip", "pxsom_comm.hip"]
# per-file extra flags: the filter works on provably finite scores (see the file header)
EXTRA_FLAGS = {"pxsom_assign_filter.hip": ["-ffinite-math-only"] + (
    ["-DSGB_VALU=" + os.environ["PXSOM_SGB_VALU"]] if "PXSOM_SGB_VALU" in os.environ else []) + (
    ["-DPXSOM_STREAM_TP=" + os.environ["PXSOM_STREAM_TP"]] if "PXSOM_STREAM_TP" in os.environ else []) + (
    ["-DPXSOM_PACKED_TMERGE=" + os.environ["PXSOM_PACKED_TMERGE"]] if "PXSOM_PACKED_TMERGE" in os.environ else [])}
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
               "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function"] + os.environ.get("PXSOM_HIPCC_EXTRA", "").split()   # (ablation builds)


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libpxsom.so cannot be built")
    return exe


HEADERS = ["pxsom_common.h", "pxsom_assign.h", "pxsom_wave.h", "pxsom_assign_filter_fast.h", "pxsom_batch_step.h", "pxsom_prep.h",
           "pxsom_sums.h", "pxsom_assign_onepass.h", "pxsom_xch.h"]
STAMP_PATH = SO_PATH + ".srchash"


def _source_digest() -> str:
    """sha256 over every input of the build (sources, headers, flags).  Content, not mtimes: a tree copied
    to another box (gpurun snapshot, rsync) keeps its prebuilt library valid whatever the copy did to the
    timestamps."""
    h = hashlib.sha256()
    files = [os.path.join(_PKG, "csrc", f) for f in SOURCES + HEADERS] + [os.path.join(_ROOT, "include", "pxsom.h")]
    for path in files:
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    h.update(repr((HIPCC_FLAGS, sorted(EXTRA_FLAGS.items()))).encode())
    return h.hexdigest()


def needs_build() -> bool:
    if not (os.path.exists(SO_PATH) and os.path.exists(STAMP_PATH)):
        return True
    with open(STAMP_PATH) as f:
        return f.read().strip() != _source_digest()


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 and link libpxsom.so.  Returns its path.
    Safe to call from several processes at once (one rank per GPU): an exclusive file lock lets one of
    them build while the others wait and then find the library up to date."""
    if not force and not needs_build():
        return SO_PATH
    with open(os.path.join(_PKG, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():
                return SO_PATH
            return _build_locked(verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(verbose: bool) -> str:
    hipcc = _hipcc()
    objdir = os.path.join(_PKG, "csrc", "_obj")
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc, *HIPCC_FLAGS, *EXTRA_FLAGS.get(src, []), "-I", os.path.join(_ROOT, "include"), "-I",
               os.path.join(_PKG, "csrc"), "-c", os.path.join(_PKG, "csrc", src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        jobs.append((src, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, _, proc in jobs:   # the translation units compile side by side
        out, _ = proc.communicate()
        if proc.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode(errors='replace')}")
        if verbose and out:
            print(out.decode(errors="replace"))
    tmp = SO_PATH + ".tmp"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *[o for _, o, _ in jobs], "-ldl", "-o", tmp])
    os.replace(tmp, SO_PATH)
    with open(STAMP_PATH + ".tmp", "w") as f:
        f.write(_source_digest() + "\n")
    os.replace(STAMP_PATH + ".tmp", STAMP_PATH)
    return SO_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
