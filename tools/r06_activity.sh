#!/bin/bash
# round 6: the 8-way partitions on a recording with quiet and busy stretches (synth.activity_profile): contiguous against block-cyclic shards
mkdir -p gpurun_out/r06
SEED=${1:-11}
W=gpurun_out/r06/vae_cache.pt
for blk in 0 32 8; do
  python bench.py --workload configs4 --activity $SEED --block $blk --emulate-ranks 8 --steps 3 --warmup 2 --cpu-windows 0 --weights-cache $W \
      > gpurun_out/r06/activity_configs4_block${blk}.log 2> gpurun_out/r06/activity_configs4_block${blk}.err || { tail -5 gpurun_out/r06/activity_configs4_block${blk}.err; exit 1; }
  python - <<PY
import json
r=json.loads(open("gpurun_out/r06/activity_configs4_block${blk}.log").read().strip().splitlines()[-1])
json.dump(r, open("gpurun_out/r06/activity_configs4_block${blk}.json", "w"), indent=1)
p=r["partition"]
print("configs4 block ${blk}: value %.0f  evals max/mean %.3f  time max/mean %.3f  per-rank evals %s  ms %s" % (r["value"], p["evaluations_per_rank"]["max_over_mean"], p["time_per_rank"]["max_over_mean"], [q["evaluations"] for q in p["per_rank"]], [q["ms_best_step"] for q in p["per_rank"]]))
PY
done
rm -f $W
