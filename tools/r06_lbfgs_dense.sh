#!/bin/bash
# (the kernel variants this script switched between -- GEM_LBFGS_DENSE -- were NOT kept: see profiles/lbfgs_dense_experiment_r06.txt; the script stays as the record of the command)
# round 6: lbfgs_advance with the bf16 ring packed in registers, product form (103 VGPRs, four workgroups per CU) against the form held to
# 80 VGPRs (six per CU: 1536 windows resident at once), bf16, 1536 / 1563-like / 8192 windows
for rep in 1 2; do for wl in 128 w8192x; do for dense in 0 1; do
  if [ $dense = 1 ]; then export GEM_DEV=1 GEM_LBFGS_DENSE=1; else unset GEM_LBFGS_DENSE; fi
  python bench.py --workload $wl --precision bf16 --no-extra --no-partition --cpu-windows 0 --vae structured --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); rl=r['roofline'].get('lbfgs') or {}
print('rep $rep workload $wl dense $dense: %.0f windows/s  ms/step %.3f  lbfgs %s us' % (r['value'], r['ms_per_step'], rl.get('avg_us')))"
done; done; done
