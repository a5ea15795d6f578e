#!/bin/bash
# (the switch GEM_FILE_STAGE_MMAP this script used was NOT kept: see profiles/stage_mmap_experiment_r06.txt)
# round 6: page cache -> pinned staging through pread (product) against a mapping of the file + memcpy (GEM_DEV=1 GEM_FILE_STAGE_MMAP=1)
for rep in 1 2; do for m in 0 1; do
  if [ $m = 1 ]; then export GEM_DEV=1 GEM_FILE_STAGE_MMAP=1; else unset GEM_FILE_STAGE_MMAP; fi
  echo "=== rep $rep mmap $m"
  python tools/whole_sequence_timing.py structured 2>&1 | grep "end to end" | sed 's/reading.*end to end/end to end/; s/optimized_global.*//' | grep -v "2 batches"
done; done
