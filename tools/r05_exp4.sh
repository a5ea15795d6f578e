#!/bin/bash
# Runs ON THE GPU BOX: the device-wide-barrier experiment -- backward front product + lbfgs_advance as ONE launch (fp32, 240 windows)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
W="--weights-cache /tmp/vae_cache.pt --cpu-windows 0 --no-extra --no-profile --steps 50 --warmup 5"
python bench.py --weights-cache /tmp/vae_cache.pt --cpu-windows 0 --no-extra --no-profile --steps 2 --warmup 1 > /dev/null 2>&1 || exit 1
run() { name=$1; shift; v=$(timeout -k 5 200 env "$@" 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], 'us per round %.1f' % (d['ms_per_step'] * 1e3 / 64), d['mpjpe_mm'], d['evals_per_stage'])"); echo "$name: $v"; }
run "two launches (product)"          python bench.py $W
run "one launch + grid barrier"       GEM_DEV=1 GEM_FUSE_BWD_LBFGS=1 python bench.py $W
run "two launches (product)"          python bench.py $W
run "one launch + grid barrier"       GEM_DEV=1 GEM_FUSE_BWD_LBFGS=1 python bench.py $W
