"""Developer aid: the smoke() stage against the oracle, window by window, under a few developer switches (set in the environment)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from globalegomocap_amd import synth, vae as vae_schema
from globalegomocap_amd.camera import FisheyeCamera, DEFAULT_CALIBRATION
from globalegomocap_amd.engine import WindowEngine, energy_weights, LOCAL_STAGE, stats_to_numpy
from oracle import np_oracle as O

shape = vae_schema.VAEShape(latent_dim=64, hidden=(16, 16, 32, 32, 64))
sd = vae_schema.synthetic_state_dict(shape, seed=3, gain=2.0)
cam = FisheyeCamera.from_json(DEFAULT_CALIBRATION)
seq = synth.make_sequence(n_frames=26, seed=4, camera=cam)
est = np.asarray(seq["estimated_local_skeleton"], dtype=np.float32)
heat = np.asarray(seq["heatmap_list"], dtype=np.float32)
starts = np.array([0, 8, 16], dtype=np.int32)
pose = np.stack([est[s:s + 10] for s in starts])
eng = WindowEngine(shape, cam, max_windows=8)
eng.load_vae(LOCAL_STAGE, sd)
mb = eng.mean_bone_length(est)
rng = np.random.default_rng(0)
z = rng.normal(size=(3, shape.latent_dim)).astype(np.float32)
w = energy_weights(1e-2, 1e-2, 1e-1, 1e-3, 1e-2)
eps = rng.normal(size=(3, shape.latent_dim)).astype(np.float32)
out, stats = eng.optimize_stage(LOCAL_STAGE, pose, mb, eps, w, heat, starts)
torch.cuda.synchronize()
st = stats_to_numpy(stats)
tr = eng.read_trace(3)
vae = O.fold_vae(sd)
ocam = O.Camera(poly=np.asarray(cam.poly_w2c), cx=cam.cx, cy=cam.cy)
ow = O.Weights(1e-2, 1e-2, 1e-1, 1e-3, 1e-2)
mb_o = O.mean_bone_length(est)
for b in range(3):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from helpers import oracle_stage_losses
    ref, so, losses = oracle_stage_losses(vae, ocam, ow, pose[b], heat[starts[b]:starts[b] + 10], mb_o, eps[b])
    err = float(np.linalg.norm(out[b].cpu().numpy() - ref, axis=-1).mean())
    print("window", b, "err mm %.3f" % (err * 1e3), "hip n_iter/evals/loss", st["n_iter"][b], st["func_evals"][b], st["final_loss"][b],
          "oracle", so["n_iter"], so["func_evals"], so["loss"])
    t = tr[b]
    t = t[~np.isnan(t)]
    o = np.asarray(losses, dtype=np.float64)
    n = min(len(t), len(o))
    bad = np.nonzero(np.abs(t[:n] - o[:n]) > 1e-4 * np.abs(o[:n]))[0]
    print("   evaluations hip / oracle:", len(t), len(o), " first closure value off by > 1e-4 at round", (int(bad[0]) if len(bad) else None))
    if len(bad):
        k = int(bad[0])
        print("   hip   ", np.array2string(t[max(0, k - 2):k + 3], precision=10))
        print("   oracle", np.array2string(o[max(0, k - 2):k + 3], precision=10))
