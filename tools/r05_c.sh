#!/bin/bash
# Runs ON THE GPU BOX: the whole GPU suite + the default bench line
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-r05c}; mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; rc=$?; tail -15 $OUT/pytest_gpu.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 || exit 1
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --full-record $OUT/bench_full.json > $OUT/bench_default.json 2> $OUT/bench_default.err || { tail -20 $OUT/bench_default.err; exit 1; }
python - $OUT/bench_default.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(json.dumps(d["summary"], indent=None))
PY
