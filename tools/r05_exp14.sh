#!/bin/bash
# Runs ON THE GPU BOX: the bf16 front products at 13 row tiles (1537-1664 windows): K cut chosen by the 2-per-CU rule (1 for the
# forward product: 260 tiles) against a forced cut of 2 (520 workgroups on 512 slots)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
W="--weights-cache /tmp/vae_cache.pt --cpu-windows 0 --no-extra --no-profile --steps 20 --warmup 3 --precision bf16"
python bench.py --weights-cache /tmp/vae_cache.pt --cpu-windows 0 --no-extra --no-profile --steps 2 --warmup 1 > /dev/null 2>&1 || exit 1
run() { name=$1; shift; v=$(timeout -k 5 300 env "$@" 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['mpjpe_mm']['optimised'])"); echo "$name: $v"; }
for wl in 128 130 131 138 150 160; do
run "$wl chunks: default"   python bench.py $W --workload $wl
run "$wl chunks: K cut 2"   GEM_DEV=1 GEM_BF16_SK=2 python bench.py $W --workload $wl
done
