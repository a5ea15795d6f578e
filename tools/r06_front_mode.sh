#!/bin/bash
# round 6: forward front product handing the tail bf16 (no K cut) against the fp32 slab pair, 1536 / 1563 / 2052 / 768 windows bf16
mkdir -p gpurun_out/r06
for nc in 128 64 171; do for mode in 0 1 2; do
  GEM_DEV=1 GEM_FRONT_FWD_MODE=$mode python bench.py --workload $nc --precision bf16 --no-extra --no-partition --cpu-windows 0 --vae structured --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
ro=r.get('roofline') or {}; rt=r.get('roofline_tail') or {}; rl=r.get('lbfgs_advance') or {}
print('chunks $nc mode $mode: %.0f windows/s  ms/step %.3f  gemm %s us  tail %s us  lbfgs %s us' % (r['value'], r['ms_per_step'], ro.get('avg_us'), rt.get('avg_us'), rl.get('avg_us')))"
done; done
