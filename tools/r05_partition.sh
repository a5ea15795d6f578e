#!/bin/bash
# Runs ON THE GPU BOX: BASELINE configs[3] / configs[4] at FULL size under the 8-way partition, the eight shards one after the
# other on one card (bench.py --emulate-ranks 8), + a first look at the MFMA-busy counters.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r05a; mkdir -p $OUT
W="--weights-cache /tmp/vae_cache.pt --cpu-windows 0"
timeout -k 10 900 python bench.py --workload configs3 --emulate-ranks 8 --steps 3 --warmup 1 $W > $OUT/configs3_emu8.json 2> $OUT/configs3_emu8.err || { tail -20 $OUT/configs3_emu8.err; exit 1; }
cut -c1-400 $OUT/configs3_emu8.json
timeout -k 10 900 python bench.py --workload configs4 --emulate-ranks 8 --steps 3 --warmup 1 $W > $OUT/configs4_emu8.json 2> $OUT/configs4_emu8.err || { tail -20 $OUT/configs4_emu8.err; exit 1; }
cut -c1-400 $OUT/configs4_emu8.json
rocprofv3 -L > $OUT/counters.txt 2>&1
grep -i -c mfma $OUT/counters.txt
for wl in "f32 --steps 2 --warmup 1" "bf16_8192 --steps 1 --warmup 1 --workload w8192x --precision bf16"; do
  set -- $wl; name=$1; shift
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/$name/pmc_mfma -- python bench.py "$@" --no-extra --no-profile $W > $OUT/$name.pmc.log 2>&1 || { echo FAILED $name; tail -5 $OUT/$name.pmc.log; exit 1; }
  f=$(find $OUT/$name/pmc_mfma -name '*counter_collection.csv' | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
f = sys.argv[1]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    acc[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
with open(f.replace("counter_collection.csv", "counter_summary.csv"), "w") as o:
    w = csv.writer(o); w.writerow(["kernel", "counter", "dispatches", "mean_value"])
    for (k, c), v in sorted(acc.items()):
        w.writerow([k, c, len(v), sum(v) / len(v)])
        if "gem::" in k: print(k[:70], c, len(v), sum(v) / len(v))
PY
  find $OUT/$name/pmc_mfma -name '*counter_collection.csv' -delete
done
