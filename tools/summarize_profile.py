#!/usr/bin/env python3
"""Turn a gpurun_out/<tag>p/ dump of tools/collect_profiles.sh into the small summaries kept under profiles/.

    python tools/summarize_profile.py gpurun_out/r02p profiles r02
"""
import csv
import glob
import json
import os
import sys


def newest(files):
    """gpurun merges a call's files into what earlier calls left behind: only the most recent run of a directory counts"""
    return sorted(files, key=os.path.getmtime, reverse=True)[:1]

src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
os.makedirs(dst, exist_ok=True)


def short(name):
    return name.replace("void gem::", "gem::").replace("void ", "").split("(")[0].strip()


def kernel_stats(sub, out_name):
    f = newest(glob.glob(os.path.join(src, sub, "trace", "*", "*kernel_stats.csv")))
    if not f:
        return
    rows = list(csv.DictReader(open(f[0])))
    line = [l for l in open(os.path.join(src, sub, "trace.log")) if l.startswith("{")]
    with open(os.path.join(dst, out_name), "w") as o:
        if line:
            d = json.loads(line[-1])
            o.write("# rocprofv3 --kernel-trace --stats of: python bench.py (see profiles/README.md); this run printed value = %s %s, %s ms per step, "
                    "workload: %s\n" % (d["value"], d["unit"], d["ms_per_step"], d["config"]["workload"]))
        w = csv.writer(o)
        w.writerow(["kernel", "calls", "total_ms", "avg_us", "pct", "min_us", "max_us"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], "%.3f" % (float(r["TotalDurationNs"]) / 1e6), "%.2f" % (float(r["AverageNs"]) / 1e3),
                        r["Percentage"], "%.2f" % (float(r["MinNs"]) / 1e3), "%.2f" % (float(r["MaxNs"]) / 1e3)])


def traffic(sub, dominant_substr, out_name, note_cmd, windows, precision):
    t = {}
    for key, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        f = newest(glob.glob(os.path.join(src, sub, "pmc_" + key, "*", "*counter_summary.csv")))
        if not f:
            return None
        t[key] = {short(r["kernel"]): (float(r["mean_value"]), int(r["dispatches"])) for r in csv.DictReader(open(f[0])) if r["counter"] == counter}
    out = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `%s`; counters are in KiB; on gfx950 FETCH_SIZE "
                   "reports half of a wide coalesced stream, so read bytes = 2 * FETCH_SIZE * 1024 (MI355X_MICROARCH.md section HBM); "
                   "per launch = mean over dispatches" % note_cmd,
           # what bench.py's committed_traffic() matches on: a kernel's bytes are only ever reported for THIS workload
           "workload": {"windows": windows, "precision": precision}, "kernels": {}}
    for k in sorted(set(t["fetch"]) | set(t["write"])):
        if not k.startswith("gem::"):
            continue
        fe, wr = t["fetch"].get(k, (0.0, 0)), t["write"].get(k, (0.0, 0))
        out["kernels"][k] = {"dispatches": fe[1], "fetch_kib_raw": round(fe[0], 1), "read_bytes_corrected": round(2 * fe[0] * 1024),
                             "write_bytes": round(wr[0] * 1024)}
    dom = [k for k in out["kernels"] if any(d in k for d in dominant_substr.split("|"))]
    if dom:       # (forward and backward-data may be two instantiations: dispatch-weighted mean per launch)
        n = sum(out["kernels"][k]["dispatches"] for k in dom)
        out["dominant_kernel"] = " + ".join(dom)
        out["hbm_bytes_per_launch"] = round(sum((out["kernels"][k]["read_bytes_corrected"] + out["kernels"][k]["write_bytes"]) *
                                                out["kernels"][k]["dispatches"] for k in dom) / max(n, 1))
    # third pass (round 5): SQ_VALU_MFMA_BUSY_CYCLES (MFMA-pipe busy cycles summed over the 1024 SIMDs) and GRBM_GUI_ACTIVE (summed over
    # the 8 XCDs: / 8 = the dispatch's active cycles, profiler overhead of ~15-20 k cycles included) of the same command
    f = newest(glob.glob(os.path.join(src, sub, "pmc_mfma", "*", "*counter_summary.csv")))
    if f:
        m = {}
        for r in csv.DictReader(open(f[0])):
            m.setdefault(short(r["kernel"]), {})[r["counter"]] = (float(r["mean_value"]), int(r["dispatches"]))
        out["note_mfma"] = ("rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE, a third pass of the same command: "
                            "mfma_busy_cycles = busy cycles of the matrix pipes summed over the 1024 SIMDs, per launch; gui_active_cycles = "
                            "GRBM_GUI_ACTIVE / 8 XCDs (the profiled dispatch's ~15-20 k cycles of overhead included); mfma_busy_frac_of_active "
                            "= mfma_busy_cycles / (1024 * gui_active_cycles)")
        for k, v in m.items():
            if k in out["kernels"] and "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v:
                busy, act = v["SQ_VALU_MFMA_BUSY_CYCLES"][0], v["GRBM_GUI_ACTIVE"][0] / 8.0
                out["kernels"][k].update({"mfma_dispatches": v["SQ_VALU_MFMA_BUSY_CYCLES"][1], "mfma_busy_cycles": round(busy),
                                          "gui_active_cycles": round(act), "mfma_busy_frac_of_active": round(busy / (1024.0 * act), 4) if act else None})
    json.dump(out, open(os.path.join(dst, out_name), "w"), indent=1)
    return out


kernel_stats("f32", "kernel_stats_%s.csv" % tag)
kernel_stats("f32_1536", "kernel_stats_%s_f32_1536_windows.csv" % tag)
kernel_stats("bf16_1536", "kernel_stats_%s_bf16_1536_windows.csv" % tag)
kernel_stats("bf16_8192", "kernel_stats_%s_bf16_8192_windows.csv" % tag)


def train_stats(out_name, sub="train", batch=64, timed=50):
    """rocprofv3 --kernel-trace --stats of tools/train_bench.py <batch> <timed>, summarised like the others (per-step launch counts)."""
    f = newest(glob.glob(os.path.join(src, sub, "trace", "*", "*kernel_stats.csv")))
    if not f:
        return
    rows = list(csv.DictReader(open(f[0])))
    log = [l for l in open(os.path.join(src, sub, "trace.log")) if l.startswith("B=")]
    steps = 5 + timed                             # 5 warm-up + the timed steps of the tool
    with open(os.path.join(dst, out_name), "w") as o:
        o.write("# rocprofv3 --kernel-trace --stats of: python tools/train_bench.py %d %d (%d steps in the trace); this run printed: %s\n"
                % (batch, timed, steps, log[-1].strip() if log else "?"))
        w = csv.writer(o)
        w.writerow(["kernel", "calls", "calls_per_step", "total_ms", "avg_us", "pct", "min_us", "max_us"])
        for r in rows:
            if float(r["Percentage"]) < 0.05:
                continue
            w.writerow([short(r["Name"]), r["Calls"], "%.1f" % (int(r["Calls"]) / steps), "%.3f" % (float(r["TotalDurationNs"]) / 1e6),
                        "%.2f" % (float(r["AverageNs"]) / 1e3), r["Percentage"], "%.2f" % (float(r["MinNs"]) / 1e3), "%.2f" % (float(r["MaxNs"]) / 1e3)])


train_stats("kernel_stats_%s_train_b64.csv" % tag)
train_stats("kernel_stats_%s_train_b1024.csv" % tag, "train1024", 1024, 20)
# the dominant kernel of the headline run = the kernel (all instantiations of a template counted together) with the most time
def total_ms(sub, substr):
    f = newest(glob.glob(os.path.join(src, sub, "trace", "*", "*kernel_stats.csv")))
    if not f:
        return 0.0
    return sum(float(r["TotalDurationNs"]) for r in csv.DictReader(open(f[0])) if substr in r["Name"]) / 1e6


cands = {"decoder_tail_kernel": total_ms("f32", "decoder_tail_kernel"), "rows::gemm_rows_kernel": total_ms("f32", "rows::gemm_rows_kernel")}
dom = max(cands, key=cands.get)
print("headline run, total ms per kernel family:", cands, "->", dom)
o = traffic("f32", dom, "traffic_%s.json" % tag,
            "python bench.py --steps 2 --warmup 1 --cpu-windows 0 --no-extra --no-profile", 240, "f32")
if o and "dominant_kernel" in o:
    json.dump({k: o.get(k) for k in ("note", "dominant_kernel", "hbm_bytes_per_launch")}, open(os.path.join(dst, "traffic_dominant_kernel.json"), "w"), indent=1)
    print("headline dominant kernel:", o["dominant_kernel"], o["hbm_bytes_per_launch"], "bytes per launch")
b = traffic("bf16_8192", "gemm_glds_kernel<false, 1, |gemm_big_kernel", "traffic_%s_bf16_8192_windows.json" % tag,
            "python bench.py --steps 1 --warmup 1 --workload w8192x --precision bf16 --cpu-windows 0 --no-extra --no-profile", 8192, "bf16")
c = traffic("bf16_1536", "gemm_glds_kernel<false, 1, ", "traffic_%s_bf16_1536_windows.json" % tag,
            "python bench.py --steps 2 --warmup 1 --workload 128 --precision bf16 --cpu-windows 0 --no-extra --no-profile", 1536, "bf16")
d = traffic("f32_1536", "gemm_glds_kernel<true, 1, |gemm_f32_kernel<1, ", "traffic_%s_f32_1536_windows.json" % tag,
            "python bench.py --steps 2 --warmup 1 --workload 128 --cpu-windows 0 --no-extra --no-profile", 1536, "f32")
if b:
    for k, v in b["kernels"].items():
        if "glds" in k or "big" in k or "lbfgs" in k or "energy" in k or "tail" in k:
            print(k, v)


def occupancy(sub, out_name):
    """round 6: the wave-occupancy pass (SQ_WAVES, SQ_WAVE_CYCLES, SQ_BUSY_CYCLES, SQ_WAIT_INST_ANY) per kernel, means over dispatches"""
    f = newest(glob.glob(os.path.join(src, sub, "pmc_occ", "*", "*counter_summary.csv")))
    if not f:
        return
    m = {}
    for r in csv.DictReader(open(f[0])):
        k = short(r["kernel"])
        if k.startswith("gem::"):
            m.setdefault(k, {"dispatches": int(r["dispatches"])})[r["counter"]] = round(float(r["mean_value"]))
    for k, v in m.items():
        if v.get("SQ_BUSY_CYCLES") and v.get("SQ_WAVE_CYCLES"):
            v["resident_waves_per_busy_cycle"] = round(v["SQ_WAVE_CYCLES"] / v["SQ_BUSY_CYCLES"], 2)
        if v.get("SQ_WAVE_CYCLES") and v.get("SQ_WAIT_INST_ANY") is not None:
            v["wave_cycles_waiting_frac"] = round(v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"], 3)
    json.dump({"note": "rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY (a pass of its own); counters summed over the chip's "
                       "shader engines as rocprofv3 reports them, means over a kernel's dispatches; resident_waves_per_busy_cycle = SQ_WAVE_CYCLES / "
                       "SQ_BUSY_CYCLES, wave_cycles_waiting_frac = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES", "kernels": m},
              open(os.path.join(dst, out_name), "w"), indent=1)
    for k in m:
        if "lbfgs" in k or "tail" in k:
            print(sub, k, m[k])


occupancy("bf16_1536", "occupancy_%s_bf16_1536_windows.json" % tag)
occupancy("bf16_8192", "occupancy_%s_bf16_8192_windows.json" % tag)
