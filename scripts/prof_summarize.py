#!/usr/bin/env python
"""Condense rocprofv3 CSV output (kernel trace / stats / counter collection) into small text
summaries that fit gpurun_out/ and can be committed under profiles/.

usage: prof_summarize.py <rocprof_out_dir> <summary_out_file> [kernel-name-substring ...]
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:110]


def main():
    src, dst = sys.argv[1], sys.argv[2]
    keep = sys.argv[3:] or ["bmu_", "som_online", "cluster_sums", "batch_update", "batch_step", "blur", "rownorm",
                            "quantile", "normalize"]
    lines = []
    for f in sorted(glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True)):
        lines.append(f"== kernel stats ({os.path.relpath(f, src)}): rocprofv3 --kernel-trace --stats")
        rows = list(csv.DictReader(open(f)))
        lines.append("%-100s %8s %14s %12s %12s %12s %7s" % ("Name", "Calls", "TotalNs", "AvgNs", "MinNs", "MaxNs", "Pct"))
        for r in rows[:40]:
            lines.append("%-100s %8s %14s %12s %12s %12s %7s" % (
                short(r["Name"])[:100], r["Calls"], r["TotalDurationNs"], r["AverageNs"],
                r["MinNs"], r["MaxNs"], r["Percentage"]))
    for f in sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)):
        rows = list(csv.DictReader(open(f)))
        agg = defaultdict(list)
        for r in rows:
            nm = short(r["Kernel_Name"])
            if not any(k in nm for k in keep):
                continue
            key = (nm, r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""),
                   r.get("VGPR_Count", ""), r.get("Accum_VGPR_Count", ""), r.get("SGPR_Count", ""),
                   r.get("LDS_Block_Size", ""))
            agg[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        lines.append(f"== per (kernel, grid) durations from {os.path.relpath(f, src)} [ns]")
        lines.append("%-90s %9s %5s %5s %5s %7s %6s %10s %10s %10s" % (
            "kernel", "grid", "vgpr", "agpr", "sgpr", "lds", "n", "avg", "min", "max"))
        for key, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            lines.append("%-90s %9s %5s %5s %5s %7s %6d %10.0f %10d %10d" % (
                key[0][:90], key[1], key[2], key[3], key[4], key[5], len(v), sum(v) / len(v), min(v), max(v)))
    for f in sorted(glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)):
        rows = csv.DictReader(open(f))
        agg = defaultdict(list)
        for r in rows:
            nm = short(r["Kernel_Name"])
            if not any(k in nm for k in keep):
                continue
            agg[(nm, r.get("Grid_Size", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
        lines.append(f"== counters per dispatch from {os.path.relpath(f, src)} (mean over dispatches)")
        lines.append("%-90s %9s %-28s %6s %18s" % ("kernel", "grid", "counter", "n", "mean"))
        for key, v in sorted(agg.items()):
            lines.append("%-90s %9s %-28s %6d %18.1f" % (key[0][:90], key[1], key[2], len(v), sum(v) / len(v)))
    open(dst, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
