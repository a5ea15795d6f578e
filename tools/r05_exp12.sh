#!/bin/bash
# Runs ON THE GPU BOX: A/B of two libraries on the default workload (and 1536 windows fp32): in-tree library against _ab/libgem_base.so
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
W="--weights-cache /tmp/vae_cache.pt --cpu-windows 0 --no-extra --no-profile --steps 100 --warmup 5"
python bench.py --weights-cache /tmp/vae_cache.pt --cpu-windows 0 --no-extra --no-profile --steps 2 --warmup 1 > /dev/null 2>&1 || exit 1
run() { name=$1; shift; v=$(timeout -k 5 300 env "$@" 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['mpjpe_mm']['optimised'])"); echo "$name: $v"; }
for i in 1 2; do
run "240 windows: new "   python bench.py $W
run "240 windows: base"   GEM_HIP_LIB=$GRAFT_REPO_ROOT/_ab/libgem_base.so python bench.py $W
done
run "1536 windows: new "   python bench.py $W --workload 128 --steps 10
run "1536 windows: base"   GEM_HIP_LIB=$GRAFT_REPO_ROOT/_ab/libgem_base.so python bench.py $W --workload 128 --steps 10
