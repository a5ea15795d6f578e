#!/bin/bash
# round 6: copy streams x slice size for the pickle path (tools/whole_sequence_timing.py), single sequence and three pipelined
mkdir -p gpurun_out/r06
for st in 1 2 3 8; do for sl in 4 8 32; do
  echo "=== streams $st slice ${sl}MB" >> gpurun_out/r06/ws_sweep.log
  GEM_WS_STREAMS=$st GEM_WS_SLICE_MB=$sl python tools/whole_sequence_timing.py structured 2>&1 | grep "end to end" | sed 's/optimized_global.*//' >> gpurun_out/r06/ws_sweep.log
done; done
cat gpurun_out/r06/ws_sweep.log
