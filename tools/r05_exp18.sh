#!/bin/bash
# Runs ON THE GPU BOX: fp32 beyond one sequence with the slots handed out by lbfgs_advance (in-tree) against compact_kernel (_ab base)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_hip_determinism.py tests/test_hip_full_size.py -m gpu -x -q 2>&1 | tail -3 || exit 1
W="--weights-cache /tmp/vae_cache.pt --cpu-windows 0 --no-extra --no-profile --steps 10 --warmup 2"
python bench.py --weights-cache /tmp/vae_cache.pt --cpu-windows 0 --no-extra --no-profile --steps 2 --warmup 1 > /dev/null 2>&1 || exit 1
run() { name=$1; shift; v=$(timeout -k 5 300 env "$@" 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['mpjpe_mm']['optimised'])"); echo "$name: $v"; }
for wl in 40 128; do
for i in 1 2; do
run "$wl chunks fp32: atomic slots "   python bench.py $W --workload $wl
run "$wl chunks fp32: compact_kernel"  GEM_HIP_LIB=$GRAFT_REPO_ROOT/_ab/libgem_base.so python bench.py $W --workload $wl
done
done
