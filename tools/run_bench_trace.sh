#!/bin/bash
# Runs ON THE GPU BOX: kernel trace + stats of one bench configuration.   bash tools/run_bench_trace.sh <tag> <bench args...>
set -o pipefail
TAG=$1; shift
OUT=gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p $OUT
if [ ! -f /tmp/vae_cache.pt ]; then
  python bench.py --steps 2 --warmup 1 --cpu-windows 0 --no-extra --no-profile --weights-cache /tmp/vae_cache.pt > $OUT/cache.log 2>&1 || { tail -5 $OUT/cache.log; exit 1; }
fi
python bench.py "$@" --cpu-windows 0 --no-extra --weights-cache /tmp/vae_cache.pt > $OUT/plain.log 2>&1 || { tail -5 $OUT/plain.log; exit 1; }
grep '^{' $OUT/plain.log | cut -c1-200
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py "$@" --cpu-windows 0 --no-extra --weights-cache /tmp/vae_cache.pt > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
grep '^{' $OUT/trace.log | cut -c1-200
find $OUT -name '*_kernel_trace.csv' -delete
f=$(find $OUT/trace -name '*kernel_stats.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    print("%-100s %6s %9.3f ms %8.2f us" % (r["Name"].replace("void ", "").split("(")[0][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
