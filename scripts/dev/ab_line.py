"""One bench.py line on a variant build of libpxsom.so (same-box A/B of timing builds made in the build container).

usage: python scripts/dev/ab_line.py <path/to/variant.so> [bench.py arguments]
Variants: scripts/dev/build_variant.sh <name> "<extra hipcc flags>" -> ark_analysis_amd/variants/<name>.so (git-ignored, travels
with gpurun snapshots)."""
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
