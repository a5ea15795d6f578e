#!/bin/bash
# Runs ON THE GPU BOX: the 4-wave fp32 tail with three workgroups per CU (48 KB LDS carve, 167 VGPRs)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_hip_determinism.py -m gpu -x -q 2>&1 | tail -5 || exit 1
W="--weights-cache /tmp/vae_cache.pt --cpu-windows 0 --no-extra --no-profile --steps 8 --warmup 2"
python bench.py --weights-cache /tmp/vae_cache.pt --cpu-windows 0 --no-extra --no-profile --steps 2 --warmup 1 > /dev/null 2>&1 || exit 1
run() { name=$1; shift; v=$(timeout -k 5 300 env "$@" 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['mpjpe_mm']['optimised'])"); echo "$name: $v"; }
for wl in 40 64 128 213 256 341; do
run "$wl chunks: 4-wave tail, no cap"     GEM_DEV=1 GEM_TAIL_CAP=1000 python bench.py $W --workload $wl
done
run "20 chunks (240): default"            python bench.py $W --steps 20
