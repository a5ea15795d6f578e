#!/bin/bash
# Runs ON THE GPU BOX: the 8196-window bf16 shard with one lane and with two, then a kernel trace of the two-lane run and the
# measured overlap of the L-BFGS advance with the GEMM / tail kernels.   bash tools/lanes_bench.sh
set -o pipefail
OUT=gpurun_out/lanes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p $OUT
C="--workload w8192 --precision bf16 --cpu-windows 0 --no-extra --no-profile --weights-cache /tmp/vae_cache.pt"
[ -f /tmp/vae_cache.pt ] || python bench.py --steps 2 --warmup 1 --cpu-windows 0 --no-extra --no-profile --weights-cache /tmp/vae_cache.pt > $OUT/cache.log 2>&1 || exit 1
for l in 0 4352; do
  python bench.py --steps 6 --warmup 2 --lanes $l $C > $OUT/lanes_$l.log 2>&1 || { tail -5 $OUT/lanes_$l.log; exit 1; }
  grep '^{' $OUT/lanes_$l.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lanes_min', $l, d['value'], 'windows/s', d['ms_per_step'], 'ms/step')"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py --steps 3 --warmup 1 --lanes 4352 $C > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
f=$(find $OUT/trace -name '*_kernel_trace.csv' | head -1)
python tools/overlap_from_trace.py $f lbfgs_advance gemm_glds
python tools/overlap_from_trace.py $f lbfgs_advance decoder_tail_bf16
python tools/overlap_from_trace.py $f decoder_tail_bf16 gemm_glds
python tools/overlap_from_trace.py $f decoder_tail_bf16 decoder_tail_bf16
find $OUT -name '*_kernel_trace.csv' -delete
