#!/bin/bash
# round 6: does tail_bf16.hip still need -fno-slp-vectorize?  The product library against one whose tail_bf16.hip was compiled with the
# SLP vectoriser on (packed fp32 arithmetic off in both), bf16, 1536 and 8192 windows, alternating; plus the bitwise tests on the variant
mkdir -p gpurun_out/r06
for rep in 1 2; do for wl in 128 w8192x; do for lib in product slp_on; do
  if [ $lib = product ]; then unset GEM_HIP_LIB; else export GEM_HIP_LIB=$PWD/build/ab/libgem_slp_on.so; fi
  python bench.py --workload $wl --precision bf16 --no-extra --no-partition --cpu-windows 0 --vae structured --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rep $rep workload $wl lib $lib: %.0f windows/s  ms/step %.3f' % (r['value'], r['ms_per_step']))"
done; done; done
export GEM_HIP_LIB=$PWD/build/ab/libgem_slp_on.so
python -m pytest tests/test_hip_determinism.py tests/test_hip_parity.py -q -m gpu -x -k "bitwise or row_tile or instantiations" 2>&1 | tail -3
