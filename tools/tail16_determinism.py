"""Developer check: repeated single evaluations through the bf16 multi-window tail must be bitwise identical.
    python tools/tail16_determinism.py [B] [reps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
if os.environ.get("GEM_LIB"):                  # a variant build of the library (developer A/B runs)
    from globalegomocap_amd import _capi
    _capi.LIB_PATH = os.path.abspath(os.environ["GEM_LIB"])
else:
    ge.build()
from globalegomocap_amd import synth, vae as V
from globalegomocap_amd.camera import FisheyeCamera, DEFAULT_CALIBRATION
from globalegomocap_amd.engine import WindowEngine, energy_weights

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
FULL = V.VAEShape()
cam = FisheyeCamera.from_json(DEFAULT_CALIBRATION)
sd = V.structured_state_dict(FULL, 7, feature_offset=0.0, signal_offset=1.0)
eng = WindowEngine(FULL, cam, max_windows=B)
eng.load_vae(0, sd)
eng.set_precision("bf16")
seq = synth.make_sequence_device(12000, seed=303, device=eng.device, cam_jitter=(0.3, 0.002))
rng = np.random.default_rng(303)
starts = rng.integers(0, 12000 - 10, B).astype(np.int32)
f0 = torch.as_tensor(starts, device=eng.device)
idx = f0.long()[:, None] + torch.arange(10, device=eng.device)[None]
pose = seq["est_local"][idx].contiguous()
mb = eng.mean_bone_length(seq["est_local"][:100]).reshape(1, 15).expand(B, 15).contiguous()
g = torch.Generator().manual_seed(1)
eps = torch.randn(B, 2048, generator=g).to(eng.device)
_, _, z = eng.encode(0, pose.reshape(B, 10, 45), eps)
w = energy_weights(1e-6, 1e-5, 1e-2, 0.0, float(os.environ.get("WR", "1e-2")))
ref = None
bad = 0
for r in range(reps):
    E, parts, dz, X = eng.energy_grad(0, z, pose, mb, w, seq["heat"], f0)
    torch.cuda.synchronize()
    cur = [t.clone() for t in (E, parts, dz, X)]
    if ref is None:
        ref = cur
        continue
    for name, a, b in zip(("E", "parts", "dz", "X"), ref, cur):
        if not torch.equal(a, b):
            bad += 1
            d = (a != b).reshape(B, -1).any(dim=1).nonzero().flatten().cpu().numpy()
            if bad <= 3: print("rep %d: %s differs on %d windows, first %s (mod 8: %s)" % (r, name, len(d), d[:12], d[:12] % 8))
            if name == "parts":
                k = int(d[0]); print("   parts ref", a[k].cpu().numpy(), "cur", b[k].cpu().numpy())
            if name == "X":
                k = int(d[0]); dd = (a[k] != b[k]).nonzero().cpu().numpy(); print("   X positions (t, j, c):", dd[:10].tolist())
print("mismatching comparisons:", bad)
