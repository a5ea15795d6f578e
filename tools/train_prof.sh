#!/bin/bash
# Runs ON THE GPU BOX: kernel stats of the batch-64 training step (top kernels)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/tfd; rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python tools/train_bench.py ${1:-64} 40 > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]: print(r["Name"][:64].ljust(64), r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
