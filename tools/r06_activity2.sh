#!/bin/bash
# round 6: the heterogeneous-partition test's sizes + configs3 on one activity stream (contiguous shards of 8192, and blocks of 64)
mkdir -p gpurun_out/r06
W=/tmp/vae_cache_r06.pt
run() { tag=$1; shift
  python bench.py "$@" --emulate-ranks 8 --steps 2 --warmup 1 --cpu-windows 0 --weights-cache $W > gpurun_out/r06/$tag.log 2> gpurun_out/r06/$tag.err || { tail -5 gpurun_out/r06/$tag.err; exit 1; }
  python - <<PY
import json
r=json.loads(open("gpurun_out/r06/$tag.log").read().strip().splitlines()[-1])
json.dump(r, open("gpurun_out/r06/$tag.json", "w"), indent=1)
p=r["partition"]
print("$tag: value %.0f  evals max/mean %.3f  time max/mean %.3f  per-rank evals %s  ms %s  mpjpe %.2f" % (r["value"], p["evaluations_per_rank"]["max_over_mean"], p["time_per_rank"]["max_over_mean"], [q["evaluations"] for q in p["per_rank"]], [q["ms_best_step"] for q in p["per_rank"]], r["mpjpe_optimised_mm"]))
PY
}
S=${1:-9}
run act${S}_c4_3124_contig --workload configs4 --windows 3124 --activity $S --block 0
run act${S}_c4_3124_block8 --workload configs4 --windows 3124 --activity $S
run act${S}_c4_contig --workload configs4 --activity $S --block 0
run act${S}_c4_block32 --workload configs4 --activity $S --block 32
run act${S}_c4_block8 --workload configs4 --activity $S
run act${S}_c3_contig --workload configs3 --activity $S
run act${S}_c3_block8 --workload configs3 --activity $S --block 8
rm -f $W
