#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
echo "== default"; timeout -k 5 200 python tools/r05_smoke_diag.py 2>&1 | tail -12
echo "== GEM_NO_TAIL"; GEM_DEV=1 GEM_NO_TAIL=1 timeout -k 5 200 python tools/r05_smoke_diag.py 2>&1 | tail -12
echo "== GEM_NO_COMPACT"; GEM_DEV=1 GEM_NO_COMPACT=1 timeout -k 5 200 python tools/r05_smoke_diag.py 2>&1 | tail -12
echo "== GEM_NO_TEXCACHE"; GEM_DEV=1 GEM_NO_TEXCACHE=1 timeout -k 5 200 python tools/r05_smoke_diag.py 2>&1 | tail -12
echo "== GEM_NO_FRONT"; GEM_DEV=1 GEM_NO_FRONT=1 timeout -k 5 200 python tools/r05_smoke_diag.py 2>&1 | tail -12
