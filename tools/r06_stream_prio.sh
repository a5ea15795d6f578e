#!/bin/bash
# round 6: do the reader's copy / report streams collide with the compute stream's hardware queue once other code has drawn streams
# from torch's pool?  burnt pool streams x stream priority (0 = torch's default pool, -1 = high priority: a pool of its own)
for burn in 0 1 2 3 5; do for prio in 0 -1; do
  echo "=== burnt $burn priority $prio"
  GEM_WS_BURN_STREAMS=$burn GEM_WS_PRIORITY=$prio python tools/whole_sequence_timing.py structured 2>&1 | grep "end to end" | sed 's/reading.*end to end/end to end/; s/optimized_global.*//' | grep -v "2 batches"
done; done
