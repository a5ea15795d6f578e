"""How much of one kernel family's run time overlaps another's, from a rocprofv3 --kernel-trace CSV.
   python tools/overlap_from_trace.py <kernel_trace.csv> lbfgs_advance gemm_glds"""
import csv, sys
path, fa, fb = sys.argv[1], sys.argv[2], sys.argv[3]
A, Bv = [], []
for r in csv.DictReader(open(path)):
    n = r["Kernel_Name"]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if fa in n: A.append((s, e))
    if fb in n: Bv.append((s, e))
A.sort(); Bv.sort()
tot = sum(e - s for s, e in A)
ov = 0
j = 0
for s, e in A:
    while j < len(Bv) and Bv[j][1] <= s: j += 1
    k = j
    while k < len(Bv) and Bv[k][0] < e:
        ov += max(0, min(e, Bv[k][1]) - max(s, Bv[k][0])); k += 1
print("%s: %d launches, %.3f ms in total; %.3f ms (%.1f %%) of it while a %s kernel was running" % (fa, len(A), tot / 1e6, ov / 1e6, 100.0 * ov / max(tot, 1), fb))
