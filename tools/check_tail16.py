"""Developer check of the bf16 multi-window tail (csrc/tail_bf16.hip) against the batched bf16 layers and the fp32 oracle:
    GEM_DEV=1 python tools/check_tail16.py
prints the deviations of decoded pose / energy / latent gradient for several batch sizes (ragged last workgroup included) and of
whole stages."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GEM_DEV"] = "1"
import torch
import __graft_entry__ as ge
ge.build()
from globalegomocap_amd import synth, vae as V
from globalegomocap_amd.camera import FisheyeCamera, DEFAULT_CALIBRATION
from globalegomocap_amd.engine import WindowEngine, energy_weights, stats_to_numpy
from oracle import np_oracle as O

W_ALL = (7e-3, 2e-2, 5e-2, 3e-3, 4e-2)
TINY = V.VAEShape(latent_dim=32, hidden=(16, 16, 32, 32, 64))
FULL = V.VAEShape()
cam = FisheyeCamera.from_json(DEFAULT_CALIBRATION)
ocam = O.Camera(poly=np.asarray(cam.poly_w2c), cx=cam.cx, cy=cam.cy)


def run(shape, sd, B, tail16, pose, mb, z, heat, starts, eps, stage_w):
    # tail16: True = the multi-window bf16 tail; False = the batched bf16 layers + stand-alone energy kernel (same rounding
    # points: bf16 weights and activations, fp32 accumulate; only the summation order differs)
    os.environ["GEM_TAIL16"] = "1" if tail16 else "0"
    if tail16:
        os.environ.pop("GEM_BATCHED_NARROW", None)
    else:
        os.environ["GEM_BATCHED_NARROW"] = "1"       # (same composed front layer in both runs: only the narrow layers differ)
    eng = WindowEngine(shape, cam, max_windows=B)
    eng.load_vae(0, sd)
    eng.set_precision("bf16")
    E, parts, dz, X = eng.energy_grad(0, z, pose, mb, energy_weights(*W_ALL), heat, starts)
    _, _, dz_s, _ = eng.energy_grad(0, z, pose, mb, energy_weights(*W_ALL[:4], 0.0), heat, starts)      # smooth energy: no reprojection term
    out, stats = eng.optimize_stage(0, pose, mb, eps, energy_weights(*stage_w), heat, starts)
    torch.cuda.synchronize()
    r = (E.cpu().numpy(), parts.cpu().numpy(), dz.cpu().numpy(), X.cpu().numpy(), out.cpu().numpy(), stats_to_numpy(stats), dz_s.cpu().numpy())
    eng.close()
    return r


GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
gt = np.load(os.path.join(GOLD, "lbfgs_tiny.npz"))
cases = [("tiny-golden", TINY, {k[len("local/"):]: gt[k] for k in gt.files if k.startswith("local/")}),
         ("tiny-random", TINY, V.synthetic_state_dict(TINY, 11)),
         ("full-random", FULL, V.synthetic_state_dict(FULL, 5)),
         ("full-structured", FULL, V.structured_state_dict(FULL, 7, feature_offset=0.0))]
for name, shape, sd in cases:
    vae = O.fold_vae(sd)
    for B in (3, 8, 21, 40):
        seq = synth.make_sequence(n_frames=200, seed=36)
        est = np.asarray(seq["estimated_local_skeleton"], dtype=np.float32)
        heat = np.asarray(seq["heatmap_list"], dtype=np.float32)
        rng = np.random.default_rng(B)
        starts = rng.integers(0, 190, B).astype(np.int32)
        pose = np.stack([est[s:s + 10] for s in starts])
        mb = O.mean_bone_length(est)
        mu, _ = O.encode(vae, pose.reshape(B, 10, 45))
        z = (mu + 0.1 * rng.normal(size=mu.shape)).astype(np.float32)
        eps = rng.normal(size=(B, shape.latent_dim)).astype(np.float32)
        heat_d = torch.as_tensor(heat, device="cuda")
        sw = (1e-6, 1e-5, 1e-2, 0.0, 1e-2)
        a = run(shape, sd, B, True, pose, mb, z, heat_d, starts, eps, sw)
        b = run(shape, sd, B, False, pose, mb, z, heat_d, starts, eps, sw)
        dX = np.abs(a[3] - b[3]).max() / max(1.0, np.abs(b[3]).max())
        qx = np.quantile(np.abs(a[3] - b[3]).ravel() / max(1.0, np.abs(b[3]).max()), [0.5, 0.9, 0.99, 0.999])
        qg = np.quantile(np.abs(a[2] - b[2]).ravel() / np.abs(b[2]).max(), [0.5, 0.9, 0.99, 0.999])
        cosg = [float(np.dot(a[2][k], b[2][k]) / (np.linalg.norm(a[2][k]) * np.linalg.norm(b[2][k]) + 1e-30)) for k in range(B)]
        qs = np.quantile(np.abs(a[6] - b[6]).ravel() / np.abs(b[6]).max(), [0.5, 0.9, 0.99, 0.999])
        coss = [float(np.dot(a[6][k], b[6][k]) / (np.linalg.norm(a[6][k]) * np.linalg.norm(b[6][k]) + 1e-30)) for k in range(B)]
        print("%s B=%d quantiles (50/90/99/99.9%%) of |dX|: %s   of |d dz|: %s   min cos %.6f | smooth energy: |d dz| %s min cos %.6f"
              % (name, B, np.array2string(qx, precision=2), np.array2string(qg, precision=2), min(cosg), np.array2string(qs, precision=2), min(coss)))
        dE = np.abs(a[0] - b[0]).max() / np.abs(b[0]).max()
        dP = np.abs(a[1] - b[1]).max(axis=0)
        dG = np.abs(a[2] - b[2]).max() / np.abs(b[2]).max()
        # against the fp32 oracle
        oX = oE = oG = 0.0
        for k in range(min(B, 4)):
            Xo, acts = O.decode(vae, z[k:k + 1], keep=True)
            f, p, dXo = O.energy_and_grad(Xo[0], pose[k], mb, O.Weights(*W_ALL), ocam, heat[starts[k]:starts[k] + 10])
            dzo = O.decode_backward(vae, dXo[None], acts)[0]
            oX = max(oX, np.abs(a[3][k] - Xo[0]).max() / max(1.0, np.abs(Xo[0]).max()))
            oE = max(oE, abs(a[0][k] - f) / abs(f))
            oG = max(oG, np.abs(a[2][k] - dzo).max() / np.abs(dzo).max())
        sa, sb = a[5], b[5]
        dpose = np.linalg.norm(a[4] - b[4], axis=-1).mean(axis=(1, 2)) * 1e3
        print("%s B=%d: tail16 vs batched: X %.2e E %.2e dz %.2e parts %s | vs fp32 oracle: X %.2e E %.2e dz %.2e | stage: finished %s/%s "
              "evals %.1f/%.1f loss rel %.2e pose diff mean %.3f max %.3f mm"
              % (name, B, dX, dE, dG, np.array2string(dP, precision=2), oX, oE, oG, sa["finished"].all(), sb["finished"].all(),
                 sa["func_evals"].mean(), sb["func_evals"].mean(),
                 np.abs(sa["final_loss"] - sb["final_loss"]).max() / np.abs(sb["final_loss"]).max(), dpose.mean(), dpose.max()), flush=True)
