#!/bin/bash
# Runs ON THE GPU BOX: per-kernel times at 1536 against 1560 windows, bf16
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
W="--weights-cache /tmp/vae_cache.pt --cpu-windows 0 --no-extra --steps 6 --warmup 2 --precision bf16"
python bench.py --weights-cache /tmp/vae_cache.pt --cpu-windows 0 --no-extra --no-profile --steps 2 --warmup 1 > /dev/null 2>&1 || exit 1
for wl in 128 130; do
for env in "X=1" "GEM_DEV=1 GEM_TAIL16_NRT=2"; do
echo "== $wl chunks, $env"
timeout -k 5 300 env $env python bench.py $W --workload $wl 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['value'], d['ms_per_step'], 'gemm', r.get('kernel'), r.get('avg_us'), '| tail', r['other'].get('kernel'), r['other'].get('avg_us'), '| lbfgs', r['lbfgs'].get('avg_us'))"
done
done
