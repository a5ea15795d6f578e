#!/usr/bin/env python3
"""Developer tool: time gem_trainer_step at full size (the reference's defaults: batch 64, latent 2048).
    python tools/train_bench.py [batch] [steps]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from globalegomocap_amd import synth, vae as vae_schema          # noqa: E402
from globalegomocap_amd.vae_train import VAETrainer              # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
keep = len(sys.argv) > 3 and sys.argv[3] == "keep"          # "keep": the gradients of every layer stay in the arena (p.grad semantics)
shape = vae_schema.VAEShape()
tr = VAETrainer(shape, batch_size=B, lr=1e-4)
data = torch.as_tensor(synth.make_training_windows(B, shape.seq_len, 0), device="cuda")
eps = torch.randn(B, shape.latent_dim, device="cuda")
for _ in range(5):
    tr.step(data, 0.01, eps=eps, sync=False, keep_gradients=keep)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter()
e0.record()
for _ in range(steps):
    tr.step(data, 0.01, eps=eps, sync=False, keep_gradients=keep)
e1.record()
enq = (time.perf_counter() - t0) / steps          # host time to enqueue a step (no wait for the device)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / steps
ms = e0.elapsed_time(e1) / steps
print("B=%d (%s): %.3f ms/step on the device (%.3f ms wall, %.3f ms host enqueue), %.0f windows/s, %d parameters"
      % (B, "gradients kept" if keep else "training-loop mode", ms, wall * 1e3, enq * 1e3, B / (ms * 1e-3), tr.n_params))
tr.close()
