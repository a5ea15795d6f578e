#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): HBM traffic per kernel of the training step at batch 64 -- rocprofv3 --pmc
# FETCH_SIZE and WRITE_SIZE in separate passes of `python3 tools/train_bench.py 64 10` -> gpurun_out/train_traffic.json
#   gpurun --timeout 600 -- 'bash tools/train_traffic.sh'
set -o pipefail
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/trainp
rm -rf $OUT; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $OUT/$c -- python3 $GRAFT_REPO_ROOT/tools/train_bench.py 64 10 > $OUT/$c.log 2>&1 || { echo "FAILED $c"; tail -5 $OUT/$c.log; exit 1; }
done
python3 - $OUT <<'PY'
import csv, glob, json, os, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(out, c, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                acc[r["Kernel_Name"].split("(")[0]][c].append(float(r["Counter_Value"]))
res = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of `python3 tools/train_bench.py 64 10`; counters are KiB; on gfx950 "
               "FETCH_SIZE reports half of a wide coalesced stream, so read bytes = 2 * FETCH_SIZE * 1024 (MI355X_MICROARCH.md, HBM section); "
               "per launch means", "kernels": {}}
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("FETCH_SIZE", [0]))):
    f, w = v.get("FETCH_SIZE", []), v.get("WRITE_SIZE", [])
    res["kernels"][k] = {"launches": max(len(f), len(w)), "read_MB": round(2 * 1024 * sum(f) / max(len(f), 1) / 1e6, 3),
                         "written_MB": round(1024 * sum(w) / max(len(w), 1) / 1e6, 3)}
json.dump(res, open(os.path.join(os.path.dirname(out), "train_traffic.json"), "w"), indent=1)
for k, v in list(res["kernels"].items())[:12]:
    print("%-70s %s" % (k[:70], v))
PY
rm -rf $OUT
