#!/usr/bin/env python3
"""Re-generates the round section of profiles/README.md FROM the committed summaries (kernel_stats_<tag>*.csv, traffic_<tag>*.json),
so that the numbers quoted there are the numbers in the files.

    python tools/profiles_readme.py r04        # rewrites the block between the `<!-- r04:begin -->` / `<!-- r04:end -->` markers
"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"


def stats(name, keep, top=7):
    path = os.path.join(P, name)
    if not os.path.exists(path):
        return None, []
    lines = open(path).read().splitlines()
    head = lines[0][2:] if lines[0].startswith("#") else ""
    rows = [r for r in csv.DictReader(lines[1:] if head else lines) if any(k in r["kernel"] for k in keep)]
    return head, rows[:top]


def traffic(name):
    path = os.path.join(P, name)
    return json.load(open(path)) if os.path.exists(path) else None


def mb(x):
    return "%.1f MB" % (x / 1e6)


out = []
KEEP = ("gem::",)
for title, sfile, tfile in (
        ("BASELINE configs[1]: 240 windows, fp32 (the default bench command)", "kernel_stats_%s.csv" % tag, "traffic_%s.json" % tag),
        ("BASELINE configs[2]: 1536 windows in one call, bf16", "kernel_stats_%s_bf16_1536_windows.csv" % tag, "traffic_%s_bf16_1536_windows.json" % tag),
        ("BASELINE configs[3] per-GPU shard: exactly 8192 windows, bf16, one lane", "kernel_stats_%s_bf16_8192_windows.csv" % tag,
         "traffic_%s_bf16_8192_windows.json" % tag),
        ("1536 windows, fp32", "kernel_stats_%s_f32_1536_windows.csv" % tag, "traffic_%s_f32_1536_windows.json" % tag),
        ("the training step (SURVEY 8 f.4), batch 64", "kernel_stats_%s_train_b64.csv" % tag, "traffic_%s_train_b64.json" % tag),
        ("the training step, batch 1024", "kernel_stats_%s_train_b1024.csv" % tag, None)):
    head, rows = stats(sfile, KEEP, top=9)
    if head is None:
        continue
    out.append("* **%s** -- `%s`%s" % (title, sfile, (", `%s`" % tfile) if tfile else ""))
    if "printed" in head:
        out.append("  (profiled run: %s)" % head.split("this run printed")[-1].split(", workload")[0].strip(" :"))
    tr = traffic(tfile) if tfile else None
    for r in rows:
        k = r["kernel"]
        if "compose_front" in k or "mean_bone" in k:
            continue
        extra = ""
        if tr:
            kk = tr.get("kernels", {})
            e = kk.get(k) or kk.get("void " + k)
            if e and "read_bytes_corrected" in e:
                extra = "; HBM per launch %s read + %s written" % (mb(e["read_bytes_corrected"]), mb(e["write_bytes"]))
                if e.get("mfma_busy_cycles"):
                    extra += "; MFMA pipes busy %.2f of the dispatch's active cycles (%.1f M SIMD-cycles per launch)" % (
                        e["mfma_busy_frac_of_active"], e["mfma_busy_cycles"] / 1e6)
            elif e and "read_MB" in e:
                extra = "; HBM per launch %.1f MB read + %.1f MB written" % (e["read_MB"], e["written_MB"])
        per = (" (%s per step)" % r["calls_per_step"]) if "calls_per_step" in r else ""
        out.append("  - `%s`: %s calls%s, **%s us** avg (%s .. %s), %s %% of the GPU time%s" % (k, r["calls"], per, r["avg_us"], r["min_us"], r["max_us"], r["pct"], extra))
block = "\n".join(out)
readme = os.path.join(P, "README.md")
s = open(readme).read()
b, e = "<!-- %s:begin -->" % tag, "<!-- %s:end -->" % tag
if b not in s:
    sys.exit("profiles/README.md has no %s marker" % b)
s = s[:s.index(b) + len(b)] + "\n" + block + "\n" + s[s.index(e):]
open(readme, "w").write(s)
print(block)
