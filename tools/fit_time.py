#!/usr/bin/env python3
"""Developer tool: wall time of fit_vae_device at full size (the bench's --vae fit-device recipe: 2000 steps, batch 128)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from globalegomocap_amd import synth, vae as vae_schema
from globalegomocap_amd.vae_train import fit_vae_device
shape = vae_schema.VAEShape()
win = synth.make_training_windows(4096, shape.seq_len, 0)
fit_vae_device(shape, win, steps=20, batch=128, seed=1)          # warm-up (library load, allocations)
t0 = time.perf_counter()
sd, err = fit_vae_device(shape, win, steps=2000, batch=128, seed=1)
print("fit_vae_device: 2000 steps, batch 128: %.2f s, reconstruction error %.2f mm" % (time.perf_counter() - t0, 1e3 * err))
