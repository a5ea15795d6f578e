"""Synthetic egocentric sequences in the reference's pickle schema.

The real `data/` pickles and VAE checkpoints are external downloads (`README.md:25-34`), so
tests, `bench.py` and the golden-vector script all use the workload described in SURVEY.md
section 8(d): the mean skeleton in the camera frame plus per-joint sinusoids, 2 cm estimator
noise, unit-peak Gaussian heat-maps (sigma 1.5 heat-map px) at the noise-free fisheye projection,
and a camera that translates 4 mm per frame.  The dict returned by `make_sequence` has exactly
the keys `optimizer.py:315-324` reads from `test_data.pkl`.
"""
import numpy as np

from .camera import FisheyeCamera, DEFAULT_CALIBRATION
from .skeleton import MEAN3D_MM, N_JOINTS

HEATMAP_SIZE = 64
FPS = 25.0


def rest_skeleton():
    """[15,3] metres, camera frame (z in 0.18..1.41 m in front of the head-mounted camera)."""
    return (MEAN3D_MM.T / 1000.0).copy()


def heatmap_coords(uv):
    """image pixels -> heat-map pixel coordinates sampled by optimizer.py:143-147."""
    ix = (uv[..., 0] - 128.0) * (HEATMAP_SIZE - 1) / 1024.0
    iy = uv[..., 1] * (HEATMAP_SIZE - 1) / 1024.0
    return ix, iy


def gaussian_heatmaps(ix, iy, sigma=1.5, size=HEATMAP_SIZE, dtype=np.float32):
    """[..., J] centre coordinates -> [..., size, size, J] unit-peak Gaussians (H, W, J layout)."""
    ys = np.arange(size, dtype=np.float64)[:, None, None]
    xs = np.arange(size, dtype=np.float64)[None, :, None]
    d2 = (xs - ix[..., None, None, :]) ** 2 + (ys - iy[..., None, None, :]) ** 2
    return np.exp(-d2 / (2.0 * sigma * sigma)).astype(dtype)


N_MODES = 6


def motion_modes():
    """Fixed [N_MODES,15,3] unit-peak displacement fields: the synthetic body moves as a random
    mixture of a few whole-body modes, so joint trajectories are correlated like real motion."""
    rng = np.random.default_rng(20211011)
    m = rng.normal(size=(N_MODES, N_JOINTS, 3))
    # extremities move more than the torso
    reach = np.linalg.norm(rest_skeleton() - rest_skeleton()[0], axis=1)
    m *= (0.3 + reach / reach.max())[None, :, None]
    return m / np.abs(m).max(axis=(1, 2), keepdims=True)


def make_motion(n_frames, rng, amp=0.06, t0=0.0):
    """Noise-free local poses [n,15,3]: rest skeleton + sinusoidal mixture of `motion_modes()`
    (0.3-2 Hz at 25 fps, peak displacement <= amp metres per mode)."""
    t = t0 + np.arange(n_frames, dtype=np.float64)[:, None] / FPS
    freq = rng.uniform(0.3, 2.0, size=(1, N_MODES))
    phase = rng.uniform(0.0, 2.0 * np.pi, size=(1, N_MODES))
    a = rng.uniform(0.2 * amp, amp, size=(1, N_MODES))
    coef = a * np.sin(2.0 * np.pi * freq * t + phase)                 # [n, N_MODES]
    return rest_skeleton()[None] + np.einsum("nm,mjc->njc", coef, motion_modes())


def make_cameras(n_frames, step=0.004):
    cams = np.tile(np.eye(4, dtype=np.float64), (n_frames, 1, 1))
    cams[:, 0, 3] = step * np.arange(n_frames)
    return cams


def jitter_cameras(cams, rng, rot_deg, trans_m, scale=None):
    """SLAM-like estimation noise: an independent small rotation (axis-angle, sigma rot_deg) and
    translation offset (sigma trans_m) per frame.  The optimiser is given these noisy cameras while the
    ground truth uses the true ones, like OpenVSLAM trajectories vs mocap in the reference's data.
    scale [n]: per-frame factor on both sigmas (an `activity_profile`)."""
    from scipy.spatial.transform import Rotation
    n = cams.shape[0]
    out = cams.copy()
    k = np.ones(n) if scale is None else np.asarray(scale, dtype=np.float64)
    R = Rotation.from_rotvec(rng.normal(0.0, np.deg2rad(rot_deg), size=(n, 3)) * k[:, None]).as_matrix()
    out[:, :3, :3] = R @ cams[:, :3, :3]
    out[:, :3, 3] = cams[:, :3, 3] + rng.normal(0.0, trans_m, size=(n, 3)) * k[:, None]
    return out


def activity_profile(n_frames, seed, stretch=(500, 5000), ramp=25, quiet_first=None):
    """[n_frames] in [0, 1]: a recording that alternates between QUIET stretches (0: the wearer stands still, the estimator and the
    SLAM trajectory are nearly noise-free -- the optimiser's L-BFGS leaves such windows after a few evaluations,
    /root/reference/optimizer.py:261-270) and BUSY ones (1: full motion amplitude and noise -- the stages run to their evaluation
    limit), each `stretch[0] .. stretch[1]` frames long (uniform), with linear ramps of `ramp` frames between them.  What a
    contiguous shard of a real recording looks like to one GPU: mostly one or the other."""
    rng = np.random.default_rng(seed)
    level = np.empty(n_frames)
    at, busy = 0, bool(rng.integers(2)) if quiet_first is None else not quiet_first
    while at < n_frames:
        n = int(rng.integers(stretch[0], stretch[1] + 1))
        level[at:at + n] = 1.0 if busy else 0.0
        at, busy = at + n, not busy
    if ramp > 1:
        k = np.ones(ramp) / ramp
        level = np.convolve(np.pad(level, (ramp // 2, ramp - 1 - ramp // 2), mode="edge"), k, mode="valid")
    return level


def make_sequence(n_frames=100, seed=1, camera=None, noise=0.02, sigma=1.5, with_heatmaps=True, cam_jitter=None, activity=None,
                  quiet_level=0.02):
    """Synthetic `test_data.pkl` content (numpy float64 poses/cams, float32 heat-maps).
    cam_jitter=(rot_deg, trans_m) adds SLAM-like noise to `camera_pose_list` (not to the ground truth).
    activity [n_frames] in [0, 1] (`activity_profile`): motion amplitude, estimator noise and camera noise of every frame are
    scaled by quiet_level + (1 - quiet_level) * activity, and the heat-maps' peak by the activity itself -- in a quiet stretch
    the wearer stands still and the 2-D detector sees nothing (blank heat-maps: the reprojection term of optimizer.py:139-149
    contributes nothing there and the local stage's L-BFGS leaves after a few evaluations), next to busy stretches that use
    every evaluation.  The per-frame peak is returned as `heatmap_scale` (not in the reference pickle)."""
    cam = camera or FisheyeCamera.from_json(DEFAULT_CALIBRATION)
    rng = np.random.default_rng(seed)
    clean = make_motion(n_frames, rng)
    level = None
    if activity is not None:
        level = quiet_level + (1.0 - quiet_level) * np.asarray(activity, dtype=np.float64)[:n_frames]
        clean = rest_skeleton()[None] + level[:, None, None] * (clean - rest_skeleton()[None])
    est = clean + rng.normal(0.0, noise, size=clean.shape) * (1.0 if level is None else level[:, None, None])
    cams = make_cameras(n_frames)
    homo = np.concatenate([clean, np.ones(clean.shape[:2] + (1,))], axis=-1)
    gt_global = np.einsum("nij,nkj->nki", cams, homo)[..., :3]
    if cam_jitter is not None:
        cams = jitter_cameras(cams, rng, *cam_jitter, scale=level)
    out = {
        "estimated_local_skeleton": [p for p in est],
        "gt_global_skeleton": [p for p in gt_global],
        "camera_pose_list": [c for c in cams],
    }
    uv = cam.project_numpy(clean.reshape(-1, 3)).reshape(n_frames, N_JOINTS, 2)
    ix, iy = heatmap_coords(uv)
    out["heatmap_centres"] = np.stack([ix, iy], axis=-1)          # not in the reference pickle
    out["heatmap_scale"] = np.ones(n_frames) if activity is None else np.asarray(activity, dtype=np.float64)[:n_frames].copy()
    if with_heatmaps:
        hm = gaussian_heatmaps(ix, iy, sigma=sigma)
        if activity is not None:
            hm = hm * out["heatmap_scale"][:, None, None, None].astype(np.float32)
        out["heatmap_list"] = [h for h in hm]
    return out


def reference_pickle_dict(seq, frames=slice(None), heat_dtype=np.float32):
    """The dict `MakeDataForOptimization/process_test_data.py:149-157` pickles for one chunk, from a `make_sequence` dict (or
    any mapping with its keys): all FIVE keys in the writer's order, every heat-map a separate [H,W,J] array in FORTRAN order
    as `scipy.io.loadmat(...)['heatmap']` returns it (:65-67; float32 from the network's .mat files, float64 if they were
    saved as MATLAB doubles).  `pickle.dump(reference_pickle_dict(...), f)` -- default protocol, like the writer -- is a
    `test_data.pkl` as the reference's tool chain makes it."""
    est = np.asarray(seq["estimated_local_skeleton"], dtype=np.float64)[frames]
    cams = np.asarray(seq["camera_pose_list"], dtype=np.float64)[frames]
    homo = np.concatenate([est, np.ones(est.shape[:2] + (1,))], axis=-1)
    return {
        "gt_global_skeleton": [p for p in np.asarray(seq["gt_global_skeleton"], dtype=np.float64)[frames]],
        "estimated_global_skeleton": [p for p in np.einsum("nij,nkj->nki", cams, homo)[..., :3]],
        "estimated_local_skeleton": [p for p in est],
        "camera_pose_list": [c for c in cams],
        "heatmap_list": [np.asfortranarray(h, dtype=heat_dtype) for h in np.asarray(seq["heatmap_list"])[frames]],
    }


def make_training_windows(n_windows, seq_len, seed):
    """Smooth synthetic motion windows [n, seq_len, 45] for briefly fitting a test VAE."""
    rng = np.random.default_rng(seed)
    out = np.empty((n_windows, seq_len, N_JOINTS * 3), dtype=np.float32)
    for i in range(n_windows):
        out[i] = make_motion(seq_len, rng, t0=rng.uniform(0.0, 10.0)).reshape(seq_len, -1)
    return out


def make_sequence_device(n_frames, seed, device, camera=None, noise=0.02, sigma=1.5, block=200, cam_jitter=None):
    """Same content as `make_sequence`, with the big arrays created directly in HBM (torch is used as
    an allocator / elementwise engine here; this is input synthesis, not the measured path).

    Returns dict: est_local f32 [F,15,3], cams f64 [F,4,4], heat f32 [F,64,64,15] (device tensors) and
    gt_global / est_local_np / cams_np (host float64) for the metrics."""
    import torch
    seq = make_sequence(n_frames, seed, camera, noise, sigma, with_heatmaps=False, cam_jitter=cam_jitter)
    est = np.asarray(seq["estimated_local_skeleton"])
    cams = np.asarray(seq["camera_pose_list"])
    cen = torch.as_tensor(seq["heatmap_centres"], dtype=torch.float32, device=device)      # [F,15,2]
    heat = torch.empty(n_frames, HEATMAP_SIZE, HEATMAP_SIZE, N_JOINTS, dtype=torch.float32, device=device)
    ys = torch.arange(HEATMAP_SIZE, dtype=torch.float32, device=device)[None, :, None, None]
    xs = torch.arange(HEATMAP_SIZE, dtype=torch.float32, device=device)[None, None, :, None]
    for a in range(0, n_frames, block):
        c = cen[a:a + block]
        d2 = (xs - c[:, None, None, :, 0]) ** 2 + (ys - c[:, None, None, :, 1]) ** 2
        heat[a:a + block] = torch.exp(-d2 / (2.0 * sigma * sigma))
    return {
        "est_local": torch.as_tensor(est, dtype=torch.float32, device=device).contiguous(),
        "cams": torch.as_tensor(cams, dtype=torch.float64, device=device).contiguous(),
        "heat": heat,
        "gt_global": np.asarray(seq["gt_global_skeleton"]),
        "est_local_np": est,
        "cams_np": cams,
    }


def make_stream_device(n_frames, seed, device, runs=None, camera=None, noise=0.02, sigma=1.5, block=200, cam_jitter=None, host=None):
    """One long sequence of which this process holds only the frames of `runs` (list of [f0, f1); None = all) in HBM: the small
    per-frame arrays (poses, cameras, ground truth) are synthesised for the whole stream on the host (they are a function of
    the seed alone, so every rank sees the same stream), the heat-maps -- 99 % of the bytes -- only for the held frames, on the
    device.  Returns est_local / cams / heat for the held frames (concatenated runs, device) plus gt_global (host, all frames).
    `host`: the `make_sequence(n_frames, seed, ..., with_heatmaps=False)` dict of this stream when the caller already has it."""
    import torch
    seq = host if host is not None else make_sequence(n_frames, seed, camera, noise, sigma, with_heatmaps=False, cam_jitter=cam_jitter)
    keep = np.arange(n_frames) if runs is None else np.concatenate([np.arange(f0, f1) for f0, f1 in runs])
    est = np.asarray(seq["estimated_local_skeleton"])[keep]
    cams = np.asarray(seq["camera_pose_list"])[keep]
    cen = torch.as_tensor(seq["heatmap_centres"][keep], dtype=torch.float32, device=device)
    n = len(keep)
    heat = torch.empty(n, HEATMAP_SIZE, HEATMAP_SIZE, N_JOINTS, dtype=torch.float32, device=device)
    ys = torch.arange(HEATMAP_SIZE, dtype=torch.float32, device=device)[None, :, None, None]
    xs = torch.arange(HEATMAP_SIZE, dtype=torch.float32, device=device)[None, None, :, None]
    peak = torch.as_tensor(np.asarray(seq.get("heatmap_scale", np.ones(n_frames)))[keep], dtype=torch.float32, device=device)
    for a in range(0, n, block):
        c = cen[a:a + block]
        d2 = (xs - c[:, None, None, :, 0]) ** 2 + (ys - c[:, None, None, :, 1]) ** 2
        heat[a:a + block] = torch.exp(-d2 / (2.0 * sigma * sigma)) * peak[a:a + block, None, None, None]
    return {"est_local": torch.as_tensor(est, dtype=torch.float32, device=device).contiguous(),
            "cams": torch.as_tensor(cams, dtype=torch.float64, device=device).contiguous(), "heat": heat,
            "gt_global": np.asarray(seq["gt_global_skeleton"]), "est_all_np": np.asarray(seq["estimated_local_skeleton"]), "frames": keep}
