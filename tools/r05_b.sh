#!/bin/bash
# Runs ON THE GPU BOX: the new partition tests + the default bench line
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r05b; mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests/test_dist_gpu.py -x -q -m gpu > $OUT/pytest_dist.log 2>&1; rc=$?; tail -5 $OUT/pytest_dist.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --full-record $OUT/bench_full.json > $OUT/bench_default.json 2> $OUT/bench_default.err || { tail -20 $OUT/bench_default.err; exit 1; }
wc -c $OUT/bench_default.json; tail -c 2000 $OUT/bench_default.json
