#!/bin/bash
# Runs ON THE GPU BOX: two lanes running FREE (no half-round event lock) against one lane, bf16
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
W="--weights-cache /tmp/vae_cache.pt --cpu-windows 0 --no-extra --no-profile --steps 20 --warmup 3 --precision bf16"
python bench.py --weights-cache /tmp/vae_cache.pt --cpu-windows 0 --no-extra --no-profile --steps 2 --warmup 1 > /dev/null 2>&1 || exit 1
run() { name=$1; shift; v=$(timeout -k 5 200 env "$@" 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['mpjpe_mm']['optimised'])"); echo "$name: $v"; }
run "1536 one lane"           python bench.py $W --workload 128
run "1536 two lanes locked"   python bench.py $W --workload 128 --lanes 700
run "1536 two lanes free"     GEM_DEV=1 GEM_LANES_FREE=1 python bench.py $W --workload 128 --lanes 700
run "3072 one lane"           python bench.py $W --workload 256 --steps 10
run "3072 two lanes free"     GEM_DEV=1 GEM_LANES_FREE=1 python bench.py $W --workload 256 --steps 10 --lanes 1500
run "8192 one lane"           python bench.py $W --workload w8192x --steps 8
run "8192 two lanes free"     GEM_DEV=1 GEM_LANES_FREE=1 python bench.py $W --workload w8192x --steps 8 --lanes 4096
run "240 two lanes free"      GEM_DEV=1 GEM_LANES_FREE=1 python bench.py $W --lanes 100
