#!/usr/bin/env python3
"""stdin: device asm; stdout: the same with `s_waitcnt vmcnt(0) lgkmcnt(0)` after every memory instruction whose line (counted from
the kernel's label = line 1) lies in [lo, hi) of kernel <symbol substring>.  classes: comma list of vm,ds,sm (default all)."""
import re
import sys
sym, lo, hi = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
classes = set((sys.argv[4] if len(sys.argv) > 4 else "vm,ds,sm").split(","))
pat = []
if "vm" in classes:
    pat.append(r"(global|scratch|buffer|flat)_(load|store|atomic)")
if "ds" in classes:
    pat.append(r"ds_")
if "sm" in classes:
    pat.append(r"s_(buffer_)?load")
MEM = re.compile(r"^\s*(" + "|".join(pat) + ")")
inside, n = False, 0
for line in sys.stdin:
    sys.stdout.write(line)
    if not inside:
        if line.startswith("_Z") and sym in line and line.split(";")[0].rstrip().endswith(":"):
            inside, n = True, 1
        continue
    n += 1
    if "s_endpgm" in line:
        inside = False
    elif lo <= n < hi and MEM.match(line):
        sys.stdout.write("\ts_waitcnt vmcnt(0) lgkmcnt(0)\n")
