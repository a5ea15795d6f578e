#!/bin/bash
# Runs ON THE GPU BOX: 8-wave fp32 tail with a three-deep operand ring (weights two blocks ahead) against the two-deep one (A/B libraries)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_hip_determinism.py -m gpu -x -q 2>&1 | tail -3 || exit 1
W="--weights-cache /tmp/vae_cache.pt --cpu-windows 0 --no-extra --no-profile --steps 100 --warmup 5"
python bench.py --weights-cache /tmp/vae_cache.pt --cpu-windows 0 --no-extra --no-profile --steps 2 --warmup 1 > /dev/null 2>&1 || exit 1
run() { name=$1; shift; v=$(timeout -k 5 300 env "$@" 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['mpjpe_mm']['optimised'])"); echo "$name: $v"; }
for i in 1 2; do
run "240 windows: depth 3"   python bench.py $W
run "240 windows: depth 2"   GEM_HIP_LIB=$GRAFT_REPO_ROOT/_ab/libgem_depth2.so python bench.py $W
done
run "120 windows: depth 3"   python bench.py $W --workload 10
run "120 windows: depth 2"   GEM_HIP_LIB=$GRAFT_REPO_ROOT/_ab/libgem_depth2.so python bench.py $W --workload 10
