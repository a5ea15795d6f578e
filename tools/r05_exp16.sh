#!/bin/bash
# Runs ON THE GPU BOX: one 8192-window shard of configs3, eager launches against hipGraph replay
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
W="--weights-cache /tmp/vae_cache.pt --cpu-windows 0 --workload configs3 --windows 8192 --steps 8 --warmup 3"
python bench.py --weights-cache /tmp/vae_cache.pt --cpu-windows 0 --no-extra --no-profile --steps 2 --warmup 1 > /dev/null 2>&1 || exit 1
run() { name=$1; shift; v=$(timeout -k 5 300 env "$@" 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"); echo "$name: $v"; }
for i in 1 2; do
run "8192 windows: eager"   python bench.py $W
run "8192 windows: graph"   python bench.py $W --graphs
done
