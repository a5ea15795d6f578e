#!/bin/bash
# Runs ON THE GPU BOX: the fp32 fused tail as 4-wave workgroups, two per CU (tail.hip, W = 4), against the 8-wave shape / the batched
# narrow layers, at more windows than CUs
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_hip_determinism.py -m gpu -x -q 2>&1 | tail -5 || exit 1
W="--weights-cache /tmp/vae_cache.pt --cpu-windows 0 --no-extra --no-profile --steps 8 --warmup 2"
python bench.py --weights-cache /tmp/vae_cache.pt --cpu-windows 0 --no-extra --no-profile --steps 2 --warmup 1 > /dev/null 2>&1 || exit 1
run() { name=$1; shift; v=$(timeout -k 5 300 env "$@" 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['mpjpe_mm']['optimised'])"); echo "$name: $v"; }
for wl in 40 128 341; do
run "$wl chunks: default"                 python bench.py $W --workload $wl
run "$wl chunks: 8-wave tail, cap 5"      GEM_DEV=1 GEM_TAIL_WAVES=8 python bench.py $W --workload $wl
run "$wl chunks: 4-wave tail, no cap"     GEM_DEV=1 GEM_TAIL_CAP=1000 python bench.py $W --workload $wl
run "$wl chunks: 8-wave tail, no cap"     GEM_DEV=1 GEM_TAIL_WAVES=8 GEM_TAIL_CAP=1000 python bench.py $W --workload $wl
run "$wl chunks: batched narrow layers"   GEM_DEV=1 GEM_TAIL_CAP=0 python bench.py $W --workload $wl
done
run "20 chunks (240): default"            python bench.py $W --steps 20
run "20 chunks (240): 4-wave"             GEM_DEV=1 GEM_TAIL_WAVES=4 python bench.py $W --steps 20
