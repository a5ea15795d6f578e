#!/bin/bash
# Runs ON THE GPU BOX: the round's closing evidence -- noise floor of `value` (three runs of 200 timed steps), the two full-size
# 8-way partitions, the default bench line with its full record.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r05f; mkdir -p $OUT
W="--weights-cache /tmp/vae_cache.pt --cpu-windows 0"
timeout -k 10 300 python -m pytest tests/test_hip_full_size.py -x -q -k "device_wide_barrier" 2>&1 | tail -2 || exit 1
for i in 1 2 3; do timeout -k 10 300 python bench.py $W --no-extra --no-profile --steps 200 --warmup 10 2>/dev/null | grep '^{' > $OUT/noise_$i.json || exit 1; python -c "import json; d=json.load(open('$OUT/noise_$i.json')); print('noise run $i:', d['value'], d['ms_per_step'])"; done
timeout -k 10 600 python bench.py --workload configs3 --emulate-ranks 8 --steps 3 --warmup 2 $W 2>/dev/null | grep '^{' > $OUT/partition8_configs3.json || exit 1
timeout -k 10 600 python bench.py --workload configs4 --emulate-ranks 8 --steps 3 --warmup 2 $W 2>/dev/null | grep '^{' > $OUT/partition8_configs4.json || exit 1
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --full-record $OUT/bench_full.json > $OUT/bench_default.json 2> $OUT/bench_default.err || { tail -5 $OUT/bench_default.err; exit 1; }
python - $OUT/bench_default.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(json.dumps(d["summary"]))
print("roofline.mfma_busy:", d["roofline"].get("mfma_busy"), "path:", d["roofline"]["path"])
PY
