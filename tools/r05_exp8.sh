#!/bin/bash
# Runs ON THE GPU BOX: where the 4-wave fused fp32 tail stops paying against the batched narrow layers
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
W="--weights-cache /tmp/vae_cache.pt --cpu-windows 0 --no-extra --no-profile --steps 8 --warmup 2"
python bench.py --weights-cache /tmp/vae_cache.pt --cpu-windows 0 --no-extra --no-profile --steps 2 --warmup 1 > /dev/null 2>&1 || exit 1
run() { name=$1; shift; v=$(timeout -k 5 300 env "$@" 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['mpjpe_mm']['optimised'])"); echo "$name: $v"; }
for wl in 64 96 170 213 256; do
run "$wl chunks: 4-wave tail, no cap"     GEM_DEV=1 GEM_TAIL_CAP=1000 python bench.py $W --workload $wl
run "$wl chunks: batched narrow layers"   GEM_DEV=1 GEM_TAIL_CAP=0 python bench.py $W --workload $wl
done
