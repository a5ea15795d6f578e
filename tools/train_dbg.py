#!/usr/bin/env python3
"""Developer tool: three full-size steps in both update modes (and, "noisy", in the default mode twice with the first step's
parameters perturbed by one ulp); prints the relative 2-norm distance of every tensor of the parameter / moment arenas."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from globalegomocap_amd import synth, vae as vae_schema
from globalegomocap_amd.vae_train import VAETrainer, initial_state_dict, unpack_arena
B = 64
FULL = vae_schema.VAEShape()
init = initial_state_dict(FULL, 5)
poses = synth.make_training_windows(3 * B, FULL.seq_len, 4).reshape(3, B, FULL.seq_len, 45)
eps = np.random.default_rng(3).standard_normal((3, B, FULL.latent_dim)).astype(np.float32)
mode = sys.argv[1] if len(sys.argv) > 1 else "modes"
adam_eps = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-8
init_b = init
if mode == "noisy":          # the same mode twice, the second from parameters one ulp away in 1 % of the fc weights
    init_b = {k: np.array(v) for k, v in init.items()}
    w = init_b["fc_mu.weight"]; rng = np.random.default_rng(0)
    m = rng.random(w.shape) < 0.01
    w[m] = np.nextafter(w[m], np.float32(1))
a = VAETrainer(FULL, batch_size=B, lr=1e-3, weight_decay=1e-5, eps=adam_eps, state_dict=init)
b = VAETrainer(FULL, batch_size=B, lr=1e-3, weight_decay=1e-5, eps=adam_eps, state_dict=init_b)
for s in range(3):
    la = a.step(poses[s], 0.01, eps=eps[s], keep_gradients=(mode != "same2"))
    la2 = None
    lb = b.step(poses[s], 0.01, eps=eps[s], keep_gradients=(mode in ("noisy", "same")))
    print("step", s, "losses rel diff", [abs(x - y) / abs(x) for x, y in zip(la, lb)])
for what in (0, 3, 4):
    ua, ub = unpack_arena(a._down(what), FULL), unpack_arena(b._down(what), FULL)
    rows = []
    for k in ua:
        x, y = np.asarray(ua[k], np.float64), np.asarray(ub[k], np.float64)
        rows.append((np.linalg.norm(x - y) / max(1e-30, np.linalg.norm(x)), k))
    rows.sort(reverse=True)
    print("arena", what, " ".join("%s:%.1e" % (k, r) for r, k in rows[:12]))
