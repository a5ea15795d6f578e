#!/bin/bash
# Runs ON THE GPU BOX: kernel stats of the batch-64 training step (top kernels; conv_rows per grid = per layer shape)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/tfd; rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python tools/train_bench.py ${1:-64} 40 > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
python3 - <<PY
import csv,glob,collections
f=glob.glob("$OUT/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:${2:-16}]: print(r["Name"][:64].ljust(64), r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
f=glob.glob("$OUT/**/*kernel_trace.csv",recursive=True)[0]
g=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "conv_rows" in r["Kernel_Name"] or "gemm_f32_kernel" in r["Kernel_Name"] or "splitk_reduce" in r["Kernel_Name"]:
        g[(r["Kernel_Name"][:40], int(r["Grid_Size_X"])//int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]), r["Workgroup_Size_X"])].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k,v in sorted(g.items()): print(k, len(v), "avg %.2f us min %.2f" % (sum(v)/len(v)/1e3, min(v)/1e3))
PY
