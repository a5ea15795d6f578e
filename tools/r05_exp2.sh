#!/bin/bash
# Runs ON THE GPU BOX: two lanes at mid sizes (bf16), windows/s of `bench.py --no-extra`
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
W="--weights-cache /tmp/vae_cache.pt --cpu-windows 0 --no-extra --no-profile --steps 20 --warmup 3"
python bench.py $W > /dev/null 2>&1 || exit 1
run() { name=$1; shift; v=$(env "$@" 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['mpjpe_mm'])"); echo "$name: $v"; }
run "bf16 1536 one lane"        python bench.py $W --precision bf16 --workload 128
run "bf16 1536 two lanes"       python bench.py $W --precision bf16 --workload 128 --lanes 700
run "bf16 3072 one lane"        python bench.py $W --precision bf16 --workload 256 --steps 10
run "bf16 3072 two lanes"       python bench.py $W --precision bf16 --workload 256 --steps 10 --lanes 1500
run "bf16 240 one lane"         python bench.py $W --precision bf16
run "bf16 240 two lanes"        python bench.py $W --precision bf16 --lanes 100
run "bf16 8192 one lane"        python bench.py $W --precision bf16 --workload w8192x --steps 8
run "bf16 8192 two lanes"       python bench.py $W --precision bf16 --workload w8192x --steps 8 --lanes 4096
