#!/bin/bash
# Runs ON THE GPU BOX: per-kernel times of the fp32 1536-window workload with the 4-wave fused tail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
W="--weights-cache /tmp/vae_cache.pt --cpu-windows 0 --no-extra --steps 6 --warmup 2"
python bench.py --weights-cache /tmp/vae_cache.pt --cpu-windows 0 --no-extra --no-profile --steps 2 --warmup 1 > /dev/null 2>&1 || exit 1
for wl in 128 40; do
timeout -k 5 300 python bench.py $W --workload $wl 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['value'], d['ms_per_step'], {k:r.get(k) for k in ('kernel','avg_us','frac')}, 'other', r.get('other'), 'lbfgs', r.get('lbfgs'))"
done
