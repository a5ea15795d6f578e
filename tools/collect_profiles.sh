#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): rocprofv3 evidence for profiles/ -- kernel stats + HBM traffic (separate --pmc passes)
# of the default bench command (BASELINE configs[1], fp32) and of the bf16 decoder mode at configs[2] / configs[3] sizes.
#   gpurun --timeout 1200 -- 'bash tools/collect_profiles.sh r04'
set -o pipefail
TAG=${1:-r06}
OUT=gpurun_out/${TAG}p
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p $OUT
python bench.py --steps 3 --warmup 1 --cpu-windows 0 --no-extra --weights-cache /tmp/vae_cache.pt > $OUT/cache.log 2>&1 || exit 1
COMMON="--cpu-windows 0 --no-extra --weights-cache /tmp/vae_cache.pt"
run() {   # name, rocprof args, bench args
  mkdir -p $(dirname $OUT/$1); rocprofv3 $2 --output-format csv -d $OUT/$1 -- python bench.py $3 $COMMON > $OUT/$1.log 2>&1 || { echo "FAILED $1"; tail -5 $OUT/$1.log; exit 1; }
  grep '^{' $OUT/$1.log | cut -c1-160
}
# headline: configs[1] fp32
run f32/trace      "--kernel-trace --stats" "--steps 5 --warmup 1"
run f32/pmc_fetch  "--pmc FETCH_SIZE"       "--steps 2 --warmup 1 --no-profile"
run f32/pmc_write  "--pmc WRITE_SIZE"       "--steps 2 --warmup 1 --no-profile"
# third counter pass (round 5): MFMA pipe busy cycles (summed over the 1024 SIMDs) beside the active cycles of the same dispatches
run f32/pmc_mfma   "--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "--steps 2 --warmup 1 --no-profile"
# bf16 decoder mode: configs[2] (1536 windows in one call) and configs[3] per-GPU shard (8196 windows)
run bf16_1536/trace     "--kernel-trace --stats" "--steps 5 --warmup 1 --workload 128 --precision bf16"
run bf16_8192/trace     "--kernel-trace --stats" "--steps 3 --warmup 1 --workload w8192x --precision bf16"
run bf16_8192/pmc_fetch "--pmc FETCH_SIZE"       "--steps 1 --warmup 1 --workload w8192x --precision bf16 --no-profile"
run bf16_8192/pmc_write "--pmc WRITE_SIZE"       "--steps 1 --warmup 1 --workload w8192x --precision bf16 --no-profile"
run bf16_8192/pmc_mfma  "--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "--steps 1 --warmup 1 --workload w8192x --precision bf16 --no-profile"
run bf16_1536/pmc_fetch "--pmc FETCH_SIZE"       "--steps 2 --warmup 1 --workload 128 --precision bf16 --no-profile"
run bf16_1536/pmc_write "--pmc WRITE_SIZE"       "--steps 2 --warmup 1 --workload 128 --precision bf16 --no-profile"
run bf16_1536/pmc_mfma  "--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "--steps 2 --warmup 1 --workload 128 --precision bf16 --no-profile"
# fourth counter pass (round 6, bf16 only): how full the chip is and what the waves wait for -- lbfgs_advance reaches 0.68 of HBM at 8192
# windows and 0.48 at 1536 (wave counts, resident wave-cycles, busy cycles, cycles waiting for any instruction to issue)
for d in bf16_1536 bf16_8192; do
  wl=$([ $d = bf16_1536 ] && echo 128 || echo w8192x)
  mkdir -p $OUT/$d
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/$d/pmc_occ -- python bench.py --steps 1 --warmup 1 --workload $wl --precision bf16 --no-profile $COMMON > $OUT/$d/pmc_occ.log 2>&1 || echo "pmc_occ $d failed (not fatal)"
done
# fp32 at configs[2] size (the LDS-DMA fp32 kernel takes over the decoder_input products)
run f32_1536/trace      "--kernel-trace --stats" "--steps 3 --warmup 1 --workload 128"
run f32_1536/pmc_fetch  "--pmc FETCH_SIZE"       "--steps 2 --warmup 1 --workload 128 --no-profile"
run f32_1536/pmc_write  "--pmc WRITE_SIZE"       "--steps 2 --warmup 1 --workload 128 --no-profile"
run f32_1536/pmc_mfma   "--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "--steps 2 --warmup 1 --workload 128 --no-profile"
# the training step at the reference's batch of 64: kernel stats (traffic: tools/train_traffic.sh)
mkdir -p $OUT/train
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train/trace -- python tools/train_bench.py 64 50 > $OUT/train/trace.log 2>&1 || { echo "FAILED train"; tail -5 $OUT/train/trace.log; exit 1; }
grep "^B=" $OUT/train/trace.log
mkdir -p $OUT/train1024
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train1024/trace -- python tools/train_bench.py 1024 20 > $OUT/train1024/trace.log 2>&1 || { echo "FAILED train1024"; tail -5 $OUT/train1024/trace.log; exit 1; }
grep "^B=" $OUT/train1024/trace.log
# keep the summaries small: per-dispatch traces are dropped, the stats / counter tables stay
find $OUT -name '*_kernel_trace.csv' -delete
for d in f32 bf16_8192 bf16_1536 f32_1536; do
  for k in pmc_fetch pmc_write pmc_mfma pmc_occ; do
    f=$(find $OUT/$d/$k -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
f = sys.argv[1]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    acc[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
with open(f.replace("counter_collection.csv", "counter_summary.csv"), "w") as o:
    w = csv.writer(o); w.writerow(["kernel", "counter", "dispatches", "mean_value"])
    for (k, c), v in sorted(acc.items()):
        w.writerow([k, c, len(v), sum(v) / len(v)])
PY
    find $OUT/$d/$k -name '*counter_collection.csv' -delete
  done
done
du -sh $OUT
