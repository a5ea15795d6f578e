#!/bin/bash
# Runs ON THE GPU BOX: determinism of every built variant of the bf16 tail harness (two workgroups per CU, reprojection term on)
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/slp_hazard; mkdir -p $OUT
B=${1:-8192}
for v in tools/slp_hazard/t16_*; do
  n=$(basename $v)
  TAIL_DETERMINISM=1 TAIL_HEAT=1 timeout -k 5 120 $v $B > $OUT/$n.log 2>&1 || { echo "$n: FAILED rc=$?"; tail -3 $OUT/$n.log; exit 1; }
  echo "$n: $(grep determinism $OUT/$n.log) | $(grep -c 'differing values' $OUT/$n.log) bad reps | $(grep 'total inside' $OUT/$n.log)"
done
