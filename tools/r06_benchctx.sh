#!/bin/bash
# round 6: why does bench.py's host-inclusive leg measure 22 ms where the stand-alone tool measures 19.2?  the tool after bench-like
# prior activity in the process: x = a second engine ran 60 calls, p = with event profiling in the first two, m = the other precision
# modes ran too, c = that engine closed and its memory released before the timing
for ctx in "" x xp xpm xpmc; do
  echo "=== context '$ctx'"
  GEM_WS_BENCHCTX=$ctx python tools/whole_sequence_timing.py structured 2>&1 | grep "end to end" | sed 's/reading.*end to end/end to end/; s/optimized_global.*//' | grep -v "2 batches"
done
