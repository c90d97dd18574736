"""The last N kernels of a rocprofv3 kernel trace in launch order: start offset, duration, gap to the previous kernel's end, grid, name.

usage: python scripts/dev/trace_timeline.py <rocprof_out_dir> [N]"""
import csv
import glob
import os
import re
import sys

src, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 120
rows = []
for f in glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(anonymous namespace\)::|pxsom_bmu::|^void ", "", r["Kernel_Name"])
    name = re.sub(r"<.*", "", name)[:40]
    grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) // max(int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1), 1)
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print("%9.1f us  dur %8.1f  gap %6.1f  wgs %5d  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, grid, name))
    prev_end = e
