"""Developer tool: why does the global stage stop at its first test? prints latent scale and |g|^2 at z0."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from globalegomocap_amd import synth, vae as V
from globalegomocap_amd.camera import FisheyeCamera, DEFAULT_CALIBRATION
from globalegomocap_amd.engine import WindowEngine, energy_weights
from globalegomocap_amd.sequence import window_starts, relative_global_numpy, cut_windows
shape = V.VAEShape(); cam = FisheyeCamera.from_json(DEFAULT_CALIBRATION); dev = torch.device("cuda")
for kl in (0.01, 0.5):
    from globalegomocap_amd.vae_torch import fit_vae
    win = synth.make_training_windows(4096, 10, 102).reshape(-1, 10, 15, 3).copy()
    win[..., 0] += (0.004 * np.arange(10))[None, :, None]
    sd, err = fit_vae(shape, win.reshape(-1, 10, 45), kl_weight=kl, seed=102, device=dev)
    eng = WindowEngine(shape, cam, max_windows=16); eng.load_vae(1, sd)
    seq = synth.make_sequence(100, 1000, cam, with_heatmaps=False, cam_jitter=(0.3, 0.002))
    est = np.asarray(seq["estimated_local_skeleton"]); cams = np.asarray(seq["camera_pose_list"])
    starts = window_starts(100)
    clean_like = est  # noisy local as stand-in for the stage-A output
    rel = relative_global_numpy(cut_windows(est, starts), cut_windows(cams, starts)).astype(np.float32)
    mu, lv, z = eng.encode(1, rel.reshape(12, 10, 45))
    mb = eng.mean_bone_length(est.astype(np.float32))
    E, parts, dz, X = eng.energy_grad(1, z, rel, mb, energy_weights(0.01, 0.001, 0.01, 0, 0))
    print("kl %.2f recon %.2f mm | mu std %.3f  sigma mean %.4f | |X-X0| mean %.2f mm | |g|^2 %s | E %s" % (
        kl, err * 1e3, mu.std().item(), torch.exp(0.5 * lv).mean().item(),
        (X.cpu().numpy() - rel).reshape(-1, 3).__abs__().mean() * 1e3, (dz ** 2).sum(1)[:4].cpu().numpy(), E[:2].cpu().numpy()), flush=True)
