#!/bin/bash
# Runs ON THE GPU BOX: A/B measurements of round 5 (windows/s of `bench.py --no-extra`, 20 steps each)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r05exp; mkdir -p $OUT
W="--weights-cache /tmp/vae_cache.pt --cpu-windows 0 --no-extra --no-profile --steps 20 --warmup 3"
python bench.py $W > $OUT/warm.json 2>/dev/null || exit 1
run() { name=$1; shift; v=$(env "$@" 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"); echo "$name: $v"; }
run "f32 240"                     python bench.py $W
run "bf16 240 atomic slots"       python bench.py $W --precision bf16
run "bf16 240 compact kernel"     GEM_DEV=1 GEM_NO_ATOMIC_COMPACT=1 python bench.py $W --precision bf16
run "bf16 1536 atomic slots"      python bench.py $W --precision bf16 --workload 128
run "bf16 1536 compact kernel"    GEM_DEV=1 GEM_NO_ATOMIC_COMPACT=1 python bench.py $W --precision bf16 --workload 128
run "bf16 8192"                   python bench.py $W --precision bf16 --workload w8192x --steps 8
run "f32 1536 batched narrow"     python bench.py $W --workload 128 --steps 8
run "f32 1536 forced fused tail"  GEM_DEV=1 GEM_FORCE_TAIL=1 python bench.py $W --workload 128 --steps 8
run "f32 4092 batched narrow"     python bench.py $W --workload 341 --steps 4
run "f32 4092 forced fused tail"  GEM_DEV=1 GEM_FORCE_TAIL=1 python bench.py $W --workload 341 --steps 4
