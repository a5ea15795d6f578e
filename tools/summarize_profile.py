#!/usr/bin/env python3
"""Turn a gpurun_out/<round>/ rocprofv3 dump of `bench.py` into the small summaries kept under profiles/.

    python tools/summarize_profile.py gpurun_out/r01 profiles r01
"""
import collections
import csv
import glob
import json
import os
import sys

src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
os.makedirs(dst, exist_ok=True)


def short(name):
    return name.replace("void gem::", "gem::").split("(")[0]


stats = glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv"))[0]
rows = list(csv.DictReader(open(stats)))
with open(os.path.join(dst, "kernel_stats_%s.csv" % tag), "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "calls", "total_ms", "avg_us", "pct", "min_us", "max_us"])
    for r in rows:
        w.writerow([short(r["Name"]), r["Calls"], "%.3f" % (float(r["TotalDurationNs"]) / 1e6), "%.2f" % (float(r["AverageNs"]) / 1e3),
                    r["Percentage"], "%.2f" % (float(r["MinNs"]) / 1e3), "%.2f" % (float(r["MaxNs"]) / 1e3)])

traffic = {}
for key, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    f = glob.glob(os.path.join(src, "pmc_" + key, "*", "*counter_collection.csv"))
    if not f:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == counter:
            acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    traffic[key] = {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}

dom = [k for k in traffic.get("fetch", {}) if "gemm_f32_kernel<1, 0, 1, 1, 1" in k]
out = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `python bench.py --steps 2 --warmup 1 "
               "--cpu-windows 0 --no-profile`; counters are in KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced "
               "stream, so read bytes = 2 * FETCH_SIZE * 1024 (MI355X_MICROARCH.md section HBM); per launch = mean over dispatches",
       "kernels": {}}
for k in sorted(set(traffic.get("fetch", {})) | set(traffic.get("write", {}))):
    if not k.startswith("gem::"):
        continue
    fe = traffic.get("fetch", {}).get(k, (0.0, 0))
    wr = traffic.get("write", {}).get(k, (0.0, 0))
    out["kernels"][k] = {"dispatches": fe[1], "fetch_kib_raw": round(fe[0], 1), "read_bytes_corrected": round(2 * fe[0] * 1024),
                         "write_bytes": round(wr[0] * 1024)}
if dom:
    d = out["kernels"][dom[0]]
    red = [k for k in out["kernels"] if "splitk_reduce_kernel<0>" in k]
    extra = out["kernels"][red[0]] if red else {"read_bytes_corrected": 0, "write_bytes": 0}
    out["dominant_kernel"] = dom[0]
    out["hbm_bytes_per_launch"] = d["read_bytes_corrected"] + d["write_bytes"]
    out["hbm_bytes_per_launch_with_reduce"] = out["hbm_bytes_per_launch"] + extra["read_bytes_corrected"] + extra["write_bytes"]
json.dump(out, open(os.path.join(dst, "traffic_%s.json" % tag), "w"), indent=1)
json.dump({k: out.get(k) for k in ("note", "dominant_kernel", "hbm_bytes_per_launch", "hbm_bytes_per_launch_with_reduce")},
          open(os.path.join(dst, "traffic_dominant_kernel.json"), "w"), indent=1)
print(open(os.path.join(dst, "kernel_stats_%s.csv" % tag)).read()[:3000])
print(json.dumps(out, indent=1)[:3000])
