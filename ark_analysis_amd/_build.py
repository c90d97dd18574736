"""Builds libpxsom.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

The shared object sits next to this file so that it travels with the source tree (gpurun
snapshots, no site-packages install).  hipcc cross-compiles without a GPU.
"""
import os
import shutil
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
SO_PATH = os.path.join(_PKG, "libpxsom.so")
SOURCES = ["pxsom_api.hip", "pxsom_assign.hip", "pxsom_assign_filter.hip", "pxsom_train.hip",
           "pxsom_pre.hip"]
# per-file extra flags: the filter works on provably finite scores (see the file header)
EXTRA_FLAGS = {"pxsom_assign_filter.hip": ["-ffinite-math-only"] + (
    ["-DSGB_VALU=" + os.environ["PXSOM_SGB_VALU"]] if "PXSOM_SGB_VALU" in os.environ else [])}
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
               "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libpxsom.so cannot be built")
    return exe


def needs_build() -> bool:
    if not os.path.exists(SO_PATH):
        return True
    so_m = os.path.getmtime(SO_PATH)
    deps = [os.path.join(_PKG, "csrc", s) for s in SOURCES if os.path.exists(os.path.join(_PKG, "csrc", s))]
    deps += [os.path.join(_PKG, "csrc", "pxsom_common.h"), os.path.join(_PKG, "csrc", "pxsom_assign.h"),
             os.path.join(_ROOT, "include", "pxsom.h")]
    return any(os.path.getmtime(d) > so_m for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 and link libpxsom.so.  Returns its path."""
    if not force and not needs_build():
        return SO_PATH
    hipcc = _hipcc()
    objdir = os.path.join(_PKG, "csrc", "_obj")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for src in SOURCES:
        sp = os.path.join(_PKG, "csrc", src)
        if not os.path.exists(sp):
            continue
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc, *HIPCC_FLAGS, *EXTRA_FLAGS.get(src, []), "-I", os.path.join(_ROOT, "include"), "-I",
               os.path.join(_PKG, "csrc"), "-c", sp, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode(errors='replace')}")
        if verbose and out:
            print(out.decode(errors="replace"))
    tmp = SO_PATH + ".tmp"
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", tmp]
    subprocess.check_call(cmd)
    os.replace(tmp, SO_PATH)
    return SO_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
