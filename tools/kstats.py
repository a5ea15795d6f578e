#!/usr/bin/env python3
"""Print a compact per-kernel table from a rocprofv3 --kernel-trace --stats dump:  python tools/kstats.py <dir> [top]"""
import csv, glob, os, sys
src = sys.argv[1]; top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
f = sorted(glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True))
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("%-110s %7s %10s %9s %6s" % ("kernel", "calls", "total_ms", "avg_us", "pct"))
for r in rows[:top]:
    n = r["Name"].replace("void gem::", "gem::").split("(")[0]
    print("%-110s %7s %10.3f %9.2f %6.2f" % (n[:110], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
