#!/bin/bash
# Runs ON THE GPU BOX: the whole-sequence driver on pickles with the payload-skipping reader (host-inclusive rate of the default bench)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 600 python -m pytest tests/test_hip_post.py -m gpu -x -q 2>&1 | tail -3 || exit 1
timeout -k 10 600 python bench.py --steps 10 --warmup 3 --weights-cache /tmp/vae_cache.pt 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('value', d['value']); print(json.dumps(d.get('host_inclusive'))[:1500])"
