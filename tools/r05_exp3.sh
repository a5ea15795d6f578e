#!/bin/bash
# Runs ON THE GPU BOX: K cut of the front products at 1563 windows (13 row tiles of 128: 260 tiles)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
W="--weights-cache /tmp/vae_cache.pt --cpu-windows 0 --steps 10 --warmup 3 --workload configs4 --windows 1563"
python bench.py --weights-cache /tmp/vae_cache.pt --cpu-windows 0 --no-extra --no-profile --steps 2 --warmup 1 > /dev/null 2>&1 || exit 1
run() { name=$1; shift; v=$(env "$@" 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"); echo "$name: $v"; }
run "1563 default cut"  python bench.py $W
run "1563 cut 1"        GEM_DEV=1 GEM_BF16_SK=1 python bench.py $W
run "1563 cut 2"        GEM_DEV=1 GEM_BF16_SK=2 python bench.py $W
run "1563 cut 3"        GEM_DEV=1 GEM_BF16_SK=3 python bench.py $W
run "1536 default cut"  python bench.py $W --windows 1536
run "1600 default cut"  python bench.py $W --windows 1600
run "1600 cut 2"        GEM_DEV=1 GEM_BF16_SK=2 python bench.py $W --windows 1600
run "2048 default cut"  python bench.py $W --windows 2048
run "2048 cut 2"        GEM_DEV=1 GEM_BF16_SK=2 python bench.py $W --windows 2048
