#!/bin/bash
# round 6: cost of -ffp-contract=on (whole library) against round 5's flags (-ffp-contract=fast, tail_bf16.hip -fno-slp-vectorize):
# fp32 240 windows, bf16 1536 / 8192 windows, alternating
for rep in 1 2; do for cfg in "seq2k f32" "128 bf16" "w8192x bf16"; do set -- $cfg; for lib in contract_on contract_fast_noslp; do
  if [ $lib = contract_on ]; then unset GEM_HIP_LIB; else export GEM_HIP_LIB=$PWD/build/ab/libgem_contract_fast_noslp.so; fi
  python bench.py --workload $1 --precision $2 --no-extra --no-partition --cpu-windows 0 --vae structured --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rep $rep workload $1 $2 lib $lib: %.0f windows/s  ms/step %.3f' % (r['value'], r['ms_per_step']))"
done; done; done
