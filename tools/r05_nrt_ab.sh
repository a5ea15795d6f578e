#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
NRT_DUMP=/tmp/cur.npz timeout -k 10 300 python tools/r05_nrt_diag.py 2>&1 | grep "vs 2" 
GEM_LIB=build/ab/libgem_r04flags.so NRT_DUMP=/tmp/r04.npz timeout -k 10 300 python tools/r05_nrt_diag.py 2>&1 | grep "vs 2\|repeat"
python - <<'PY'
import numpy as np
a, b = np.load("/tmp/cur.npz"), np.load("/tmp/r04.npz")
for k in sorted(a.files):
    if not np.array_equal(a[k], b[k]):
        bad = np.unique(np.nonzero(a[k] != b[k])[0])
        print("current vs r04-flags build:", k, "differs on", len(bad), "windows", bad[:6], "max", np.abs(a[k] - b[k]).max())
print("compared", len(a.files), "arrays")
PY
