"""Developer check: fp32 batches between the one-sequence regime and the all-batched regime (400..1280 windows: fused tail, composed front layer in the tiled kernels) against the same windows in a small batch."""
import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from globalegomocap_amd import synth, vae as vae_schema
from globalegomocap_amd.engine import WindowEngine, energy_weights, stats_to_numpy
from globalegomocap_amd.camera import FisheyeCamera, DEFAULT_CALIBRATION
from helpers import FULL
sd = vae_schema.synthetic_state_dict(FULL, 5)
cam = FisheyeCamera.from_json(DEFAULT_CALIBRATION)
seq = synth.make_sequence(n_frames=200, seed=36)
est = np.asarray(seq["estimated_local_skeleton"], dtype=np.float32); heat = np.asarray(seq["heatmap_list"], dtype=np.float32)
rng = np.random.default_rng(10)
W_ALL = (1e-1, 1e-1, 1.0, 1e-3, 1e-2)
for B in (400, 640, 1000, 1280):
    starts = rng.integers(0, 190, B).astype(np.int32)
    pose = np.stack([est[s:s + 10] for s in starts])
    eps = rng.normal(size=(B, 2048)).astype(np.float32)
    big = WindowEngine(FULL, cam, max_windows=B); big.load_vae(0, sd)
    small = WindowEngine(FULL, cam, max_windows=40); small.load_vae(0, sd)
    mb = big.mean_bone_length(est)
    _, _, z = big.encode(0, pose.reshape(B, 10, 45), eps)
    E, parts, dz, X = big.energy_grad(0, z, pose, mb, energy_weights(*W_ALL), heat, starts)
    Es, _, dzs, Xs = small.energy_grad(0, z[:40], pose[:40], mb, energy_weights(*W_ALL), heat, starts[:40])
    out, stats = big.optimize_stage(0, pose, mb, eps, energy_weights(1e-6, 1e-5, 1e-2, 0.0, 1e-2), heat, starts)
    outs, stats_s = small.optimize_stage(0, pose[:40], mb, eps[:40], energy_weights(1e-6, 1e-5, 1e-2, 0.0, 1e-2), heat, starts[:40])
    st, ss = stats_to_numpy(stats), stats_to_numpy(stats_s)
    print("B=%d: X diff %.2e, E rel %.2e, dz rel %.2e, all done %s, evals equal %d/40, loss rel diff %.2e, pose diff %.4f mm" % (
        B, np.abs(X[:40].cpu().numpy() - Xs.cpu().numpy()).max(), np.abs(E[:40].cpu().numpy() / Es.cpu().numpy() - 1).max(),
        np.abs((dz[:40] - dzs).cpu().numpy()).max() / np.abs(dzs.cpu().numpy()).max(), st["finished"].all(),
        (st["func_evals"][:40] == ss["func_evals"]).sum(), np.abs(st["final_loss"][:40] / ss["final_loss"] - 1).max(),
        np.linalg.norm((out[:40] - outs).cpu().numpy(), axis=-1).mean() * 1e3), flush=True)
    big.close(); small.close()
