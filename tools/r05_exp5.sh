#!/bin/bash
# Runs ON THE GPU BOX: the K-cut one-round front GEMM at mid sizes (bf16), windows/s of `bench.py --no-extra`
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
W="--weights-cache /tmp/vae_cache.pt --cpu-windows 0 --no-extra --no-profile --steps 20 --warmup 3 --precision bf16"
python bench.py --weights-cache /tmp/vae_cache.pt --cpu-windows 0 --no-extra --no-profile --steps 2 --warmup 1 > /dev/null 2>&1 || exit 1
run() { name=$1; shift; v=$(timeout -k 5 200 env "$@" 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['mpjpe_mm']['optimised'])"); echo "$name: $v"; }
for wl in 128 171 256 341; do
  run "chunks $wl (x12 windows) 128x128 kernel"   python bench.py $W --workload $wl
  run "chunks $wl (x12 windows) K-cut 256x256"     GEM_DEV=1 GEM_BIG_SPLIT=1 python bench.py $W --workload $wl
done
run "1536 K-cut, 2 slices"   GEM_DEV=1 GEM_BIG_SPLIT=1 GEM_BIG_SPLIT_SK=2 python bench.py $W --workload 128
run "1536 K-cut, 8 slices"   GEM_DEV=1 GEM_BIG_SPLIT=1 GEM_BIG_SPLIT_SK=8 python bench.py $W --workload 128
run "768 windows 128x128"    python bench.py $W --workload 64
run "768 windows K-cut"      GEM_DEV=1 GEM_BIG_SPLIT=1 GEM_BIG_SPLIT_MIN=512 python bench.py $W --workload 64
