#!/bin/bash
# Runs ON THE GPU BOX: the L-BFGS advance kernel's average launch time (product library) and the mean time of every phase of the
# windows that compute a new direction from the ring (a library built with -DGEM_LB_PROBE; GEM_LBFGS_CLK=1 prints at gem_destroy).
# Build the probe library in the container first:
#   mkdir -p build/ab
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -DGEM_LB_PROBE -o build/ab/lbfgs_probe.o globalegomocap_amd/csrc/lbfgs.hip
#   hipcc --offload-arch=gfx950 -fPIC -shared -o build/ab/lib_probe.so $(ls build/obj/*.o | grep -v lbfgs.o) build/ab/lbfgs_probe.o
#   gpurun --timeout 900 -- 'bash tools/lbfgs_phase_run.sh'
export GEM_DEV=1
run() {
  for w in "" "--workload 128 --precision bf16" "--workload w8192x --precision bf16"; do
    echo "== $1 $w"
    python bench.py --steps 3 --warmup 1 --cpu-windows 0 --no-extra $w 2>&1 | grep -E "GEM_LBFGS_CLK|^\{" | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('  windows/s', d['value'], 'lbfgs avg us', (d.get('roofline_lbfgs') or {}).get('avg_us'), 'mpjpe', d.get('mpjpe_mm', {}).get('optimised'))
    else: print('  ' + l.strip()[:330])
"
  done
}
run product
GEM_LBFGS_CLK=1 GEM_HIP_LIB=$PWD/build/ab/lib_probe.so run probes
