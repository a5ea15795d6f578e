#!/bin/bash
# Runs ON THE GPU BOX: kernel stats of the 8192-window bf16 workload (top kernels) + its value
cd /tmp && export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out/c8192; rm -rf $OUT; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
python bench.py --steps 1 --warmup 0 --workload w8192x --precision bf16 --cpu-windows 0 --no-extra --weights-cache /tmp/vae_cache.pt > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python bench.py --steps 3 --warmup 1 --workload w8192x --precision bf16 --cpu-windows 0 --no-extra --weights-cache /tmp/vae_cache.pt > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if r["Name"].startswith("gem::") or "gem::" in r["Name"][:12]: print(r["Name"][:64].ljust(64), r["Calls"], r["AverageNs"])
PY
grep -o '"value": [0-9.]*' $OUT/log.txt | head -1
rm -rf $OUT/*/*kernel_trace.csv
