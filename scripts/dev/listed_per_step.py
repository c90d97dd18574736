"""Rows each fused training step lists for the exact path, on a build with -DPXSOM_STEP_COUNT_LISTED (debug counter, not in the product).

usage: python scripts/dev/listed_per_step.py <variant.so> [bench.py arguments]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ark_analysis_amd import _build   # noqa: E402

_build.SO_PATH = os.path.abspath(sys.argv[1])
_build.needs_build = lambda: False
sys.argv = ["bench.py"] + sys.argv[2:]
import bench   # noqa: E402

bench.main()
lib = ctypes.CDLL(_build.SO_PATH)
buf = (ctypes.c_uint * (3 * 4096))()
m = lib.pxsom_dbg_listed(buf, 3 * 4096)
print("launch rows listed listed/rows max_per_wg")
for i in range(m):
    r, l, mx = buf[3 * i], buf[3 * i + 1], buf[3 * i + 2]
    print(i, r, l, "%.4f" % (l / max(r, 1)), mx)
