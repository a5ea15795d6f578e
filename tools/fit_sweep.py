"""Developer tool: how many Adam steps does the synthetic VAE need? (run on the GPU box)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from globalegomocap_amd import synth, vae as V
from globalegomocap_amd.vae_torch import fit_vae
shape = V.VAEShape()
win = synth.make_training_windows(4096, 10, 101)
for seed in (1, 101, 102):
    win = synth.make_training_windows(4096, 10, seed)
    for steps, lr, kl, batch in [(2000, 2e-3, 0.01, 128), (2000, 1e-3, 0.001, 128)]:
        t = time.time()
        sd, err = fit_vae(shape, win, steps=steps, batch=batch, lr=lr, kl_weight=kl, seed=seed, device="cuda")
        print("seed %d steps %5d lr %.0e kl %.3f batch %d -> recon %.2f mm  (%.1f s)" % (seed, steps, lr, kl, batch, err * 1e3, time.time() - t), flush=True)
