"""Drop-in host interface of the hot path: `BodyPoseOptimizer` and `main`.

Same names, argument meaning and return values as the reference's `optimizer.py`
(`BodyPoseOptimizer` :33-276, `main` :311-507), with the per-window arithmetic moved into the HIP
library behind `WindowEngine`.  Differences that the reference's own interface hides:

* all windows of a chunk are optimised in ONE batched device call (the reference loops);
* the initial latent noise (`reparameterize`, SeqConvVAE.py:159-169) is drawn on the host from
  torch's global CPU generator in the reference's order (window i: local stage, then global stage),
  or passed explicitly with `eps=`;
* `smoothed_pose`, `gmm_weight`, `windows_size`, `slide_window` are accepted and ignored exactly as
  the reference ignores them; `visualization`/`save` need open3d and are refused.
"""
import os
import pickle

import numpy as np
import torch

from . import _capi
from .camera import FisheyeCamera
from .engine import WindowEngine, energy_weights, stats_to_numpy, raise_if_degenerate, LOCAL_STAGE, GLOBAL_STAGE
from .errors import calculate_errors
from .sequence import (SEQ_LEN, OVERLAP, window_starts, cut_windows, merge_batches, final_smooth,
                       relative_global_numpy, to_global_numpy)
from .skeleton import KINEMATIC_PARENTS
from .vae import infer_shape, load_checkpoint

# hard-coded in the reference's main() (optimizer.py:334,344)
GLOBAL_VAE_PATH = "networks/logs/real_full_dataset_latent_2048_len_10_slide_window_step_1_kl_0.5/checkpoints/19.pth.tar"
LOCAL_VAE_PATH = "networks/logs/only_local_full_dataset_latent_2048_len_10_kl_0.5_2/checkpoints/19.pth.tar"


def _as_state_dict(vae):
    return load_checkpoint(vae) if isinstance(vae, (str, os.PathLike)) else vae


def _raise_if_degenerate(stats):
    # a joint exactly on the optical axis makes the projection undefined; the reference raises
    # Exception("norm is zero!") from FishEyeCalibrated.py:124-127; the device latches it in bit 1 of the window status
    if stats is not None:
        raise_if_degenerate(stats)


class BodyPoseOptimizer:
    kinematic_parents = list(KINEMATIC_PARENTS)

    def __init__(self, camera_model_path, mean_skeleton, vae_path, seq_len, network_seq_len, latent_dim,
                 windows_size=5, overlap_size=1, slide_window=False, lr=2, max_iter=25, max_windows=64):
        """`vae_path` may also be an already loaded state_dict."""
        sd = _as_state_dict(vae_path)
        shape = infer_shape(sd, seq_len=network_seq_len)
        if shape.latent_dim != latent_dim:
            raise RuntimeError("size mismatch for fc_mu.weight: checkpoint latent %d, requested %d"
                               % (shape.latent_dim, latent_dim))
        self.seq_len, self.network_seq_len = seq_len, network_seq_len
        self.windows_size, self.slide_window, self.overlap_size = windows_size, slide_window, overlap_size
        self.lr, self.max_iter = lr, max_iter
        self.engine = WindowEngine(shape, FisheyeCamera.from_json(camera_model_path), max_windows=max_windows)
        self.device = self.engine.device
        self.engine.load_vae(LOCAL_STAGE, sd)      # this object owns one network: slot 0
        ms = mean_skeleton.detach().cpu().numpy() if torch.is_tensor(mean_skeleton) else np.asarray(mean_skeleton)
        self.mean_bone_length = self.engine.mean_bone_length(ms.astype(np.float32))
        self.vae_weight = self.gmm_weight = self.smooth_weight = None
        self.bone_length_weight = self.weight_3d = self.reproj_weight = None
        self.last_stats = None

    def set_weights(self, vae_weight, gmm_weight, smooth_weight, bone_length_weight, weight_3d, reproj_weight):
        self.vae_weight, self.gmm_weight, self.smooth_weight = vae_weight, gmm_weight, smooth_weight
        self.bone_length_weight, self.weight_3d, self.reproj_weight = bone_length_weight, weight_3d, reproj_weight

    def _weights(self):
        return energy_weights(self.weight_3d, self.smooth_weight, self.bone_length_weight, self.vae_weight, self.reproj_weight)

    def optimize_pose_seq_pytorch_LBFGS(self, relative_global_pose, heatmap_seq, smoothed_pose=None, eps=None):
        """One window: pose [T,15,3], heatmaps [T,H,W,15] -> float32 [T,15,3] (optimizer.py:242-276)."""
        pose = np.asarray(relative_global_pose, dtype=np.float32).reshape(1, self.seq_len, 15, 3)
        if eps is None:
            eps = torch.randn(1, self.engine.D)
        heat, frame0 = None, None
        if self.reproj_weight != 0:
            heat = np.asarray(heatmap_seq, dtype=np.float32)
            frame0 = np.zeros(1, dtype=np.int32)
        out, stats = self.engine.optimize_stage(LOCAL_STAGE, pose, self.mean_bone_length, eps, self._weights(), heat, frame0,
                                                _capi.default_lbfgs_opts(self.lr, self.max_iter))
        self.last_stats = stats_to_numpy(stats)
        _raise_if_degenerate(self.last_stats)
        return out[0].cpu().numpy()


class SequenceOptimizer:
    """Both stages of main() for every window of one or more chunks in one device call."""

    def __init__(self, camera_model_path, global_vae, local_vae, max_windows=256, lr=2, max_iter=25, seq_len=SEQ_LEN):
        sd_g, sd_l = _as_state_dict(global_vae), _as_state_dict(local_vae)
        shape = infer_shape(sd_l, seq_len=seq_len)
        if infer_shape(sd_g, seq_len=seq_len) != shape:
            raise RuntimeError("local and global VAE checkpoints have different architectures")
        self.engine = WindowEngine(shape, FisheyeCamera.from_json(camera_model_path), max_windows=max_windows)
        self.engine.load_vae(LOCAL_STAGE, sd_l)
        self.engine.load_vae(GLOBAL_STAGE, sd_g)
        self.opts = _capi.default_lbfgs_opts(lr, max_iter)
        self.seq_len = seq_len

    def stage_weights(self, vae_weight, smoothness_weight, bone_length_weight, weight_3d, reproj_weight):
        """The two `set_weights` calls of main() (optimizer.py:352-358)."""
        w_global = energy_weights(weight_3d, smoothness_weight, 0.01, vae_weight, 0.0)
        w_local = energy_weights(weight_3d / 10000, smoothness_weight / 100, bone_length_weight, vae_weight, reproj_weight)
        return w_local, w_global

    def prepare(self, est_local, cams, starts, chunk_of_window, chunk_bounds, timings=None, upload=None):
        """Checks the window tables and uploads everything a call needs EXCEPT the heat-maps and the noise (poses, cameras, first
        frames, per-chunk mean bone lengths): what can be done while the heat-maps are still on their way.  -> handle for `fire`.
        `upload(array, torch dtype) -> device tensor`: the caller's way of bringing a host array to the device (e.g. through a
        pinned block, asynchronously); default: an ordinary copy."""
        e = self.engine
        dev = e.device
        B = len(starts)
        n_frames = len(est_local)
        import time
        tick = [time.perf_counter()]

        def lap(name):          # developer timing (tools/whole_sequence_timing.py)
            if timings is not None:
                now = time.perf_counter()
                timings[name] = timings.get(name, 0.0) + (now - tick[0])
                tick[0] = now
        if len(cams) != n_frames:
            raise ValueError("est_local and cams must cover the same frames (%d / %d)" % (n_frames, len(cams)))
        if B and (int(np.min(starts)) < 0 or int(np.max(starts)) + self.seq_len > n_frames):
            raise ValueError("a window [start, start + %d) leaves the %d frames of the sequence" % (self.seq_len, n_frames))
        if len(chunk_of_window) != B or (B and int(np.max(chunk_of_window)) >= len(chunk_bounds)):
            raise ValueError("chunk_of_window must name one of the %d chunks for each of the %d windows" % (len(chunk_bounds), B))
        if upload is None:
            def upload(a, dtype):
                return torch.as_tensor(np.asarray(a), dtype=dtype).to(dev).contiguous()
        pose_d = upload(est_local, torch.float32)
        cams_d = upload(cams, torch.float64)
        lap("run: checks + upload of poses / cameras")
        mb = torch.stack([e.mean_bone_length(pose_d[a:b]) for a, b in chunk_bounds])
        mb_w = mb[upload(chunk_of_window, torch.long)].contiguous()
        f0 = upload(starts, torch.int32)
        lap("run: mean bone lengths")
        return {"pose": pose_d, "cams": cams_d, "mean_bone": mb_w, "frame0": f0, "B": B, "frames": n_frames}

    def fire(self, prep, heat, w_local, w_global, eps=None, timings=None):
        """Enqueues both stages for every window of a `prepare`d call; returns without waiting for the device (-> `collect`)."""
        e = self.engine
        dev = e.device
        B = prep["B"]
        if len(heat) != prep["frames"]:
            raise ValueError("est_local, cams and heat must cover the same frames (%d / %d)" % (prep["frames"], len(heat)))
        import time
        t0 = time.perf_counter()
        heat_d = (heat if torch.is_tensor(heat) else torch.as_tensor(np.asarray(heat))).to(dev, dtype=torch.float32).contiguous()
        if eps is None:
            eps = torch.randn(2 * B, e.D)
        eps = torch.as_tensor(np.asarray(eps) if not torch.is_tensor(eps) else eps, dtype=torch.float32).reshape(B, 2, e.D)
        eps_d = eps.to(dev, non_blocking=True)                 # one copy (asynchronous from a pinned block), split on the device
        eps_l, eps_g = eps_d[:, 0].contiguous(), eps_d[:, 1].contiguous()
        pending = e.optimize_windows(prep["pose"], prep["cams"], heat_d, prep["frame0"], prep["mean_bone"], eps_l, eps_g, w_local, w_global,
                                     self.opts)
        if timings is not None:
            timings["run: noise upload + enqueue"] = timings.get("run: noise upload + enqueue", 0.0) + time.perf_counter() - t0
        return pending

    def enqueue(self, est_local, cams, heat, starts, chunk_of_window, chunk_bounds, w_local, w_global, eps=None, timings=None):
        """`prepare` + `fire`: uploads and enqueues a whole call; returns without waiting for the device.  Arguments as `run`."""
        if len(heat) != len(est_local):
            raise ValueError("est_local, cams and heat must cover the same frames (%d / %d / %d)" % (len(est_local), len(cams), len(heat)))
        prep = self.prepare(est_local, cams, starts, chunk_of_window, chunk_bounds, timings=timings)
        return self.fire(prep, heat, w_local, w_global, eps=eps, timings=timings)

    def collect(self, pending, keep_device=False):
        """Waits for an `enqueue`d call: (mid_local, global, stats) as `run` returns them."""
        mid, glob, stats = pending
        st = stats_to_numpy(stats)
        _raise_if_degenerate(st)
        if keep_device:
            return mid, glob, st
        return mid.cpu().numpy(), glob.cpu().numpy(), st

    def run(self, est_local, cams, heat, starts, chunk_of_window, chunk_bounds, w_local, w_global, eps=None, keep_device=False, timings=None, while_device_runs=None):
        """est_local [F,15,3], cams [F,4,4], heat [F,H,W,15]; starts [B] first frame of each window;
        chunk_bounds [(f0, f1)] per chunk for the per-chunk mean bone length (optimizer.py:42-43).
        Returns (mid_local f32 [B,T,15,3], global f64 [B,T,15,3], stats); the two pose arrays stay torch device
        tensors with keep_device=True (for the device post-processing).  `while_device_runs()` is called between the enqueue
        and the blocking read-back of the statistics."""
        pending = self.enqueue(est_local, cams, heat, starts, chunk_of_window, chunk_bounds, w_local, w_global, eps=eps, timings=timings)
        if while_device_runs is not None:          # host work that does not need the result (the caller's report preparation)
            while_device_runs()
        return self.collect(pending, keep_device=keep_device)


def main(data_id, camera_model_path, vae_weight, gmm_weight, smoothness_weight, bone_length_weight, weight_3d,
         reproj_weight, visualization=False, final_smooth=False, merge=True, save=False, save_pose=False,
         global_vae_path=GLOBAL_VAE_PATH, local_vae_path=LOCAL_VAE_PATH, eps=None, optimizer=None, return_stats=False,
         device_metrics=False):
    """pickle in, poses out -- the reference's `main` (optimizer.py:311-507) for one chunk directory.

    Returns (errors OrderedDict[18], final_estimated_seq, mid_local_pose_seq, final_optimized_seq, final_gt_seq): lists of [15,3]
    frames like the reference's merge_batches, final_optimized_seq an ndarray [N',15,3] when final_smooth is True (list otherwise).
    device_metrics=True keeps the optimised windows on the device and runs the overlap merge, the Gaussian
    smoothing and `calculate_errors` there (gem_merge_windows / gem_calculate_errors) instead of in numpy.
    """
    if visualization or save:
        raise NotImplementedError("visualization/save write open3d meshes (optimizer.py:452-504): outside the hot path")
    with open("{}/test_data.pkl".format(data_id), "rb") as f:
        data = pickle.load(f)
    est_local = np.asarray(data["estimated_local_skeleton"])
    gt = np.asarray(data["gt_global_skeleton"])
    cams = np.asarray(data["camera_pose_list"])
    heat = np.asarray(data["heatmap_list"])
    seq_len, overlap = SEQ_LEN, OVERLAP
    starts = window_starts(len(est_local), seq_len, overlap)
    opt = optimizer or SequenceOptimizer(camera_model_path, global_vae_path, local_vae_path, max_windows=max(len(starts), 1))
    w_local, w_global = opt.stage_weights(vae_weight, smoothness_weight, bone_length_weight, weight_3d, reproj_weight)
    mid_local, opt_global, stats = opt.run(est_local, cams, heat, starts, np.zeros(len(starts), dtype=np.int64),
                                           [(0, len(est_local))], w_local, w_global, eps=eps, keep_device=device_metrics)
    opt_global_d = None
    if device_metrics:
        opt_global_d, mid_local = opt_global, mid_local.cpu().numpy()
    # the sequences main() returns besides the optimised one (host float64, as in the reference)
    loc_w, cam_w = cut_windows(est_local, starts, seq_len), cut_windows(cams, starts, seq_len)
    est_global = to_global_numpy(relative_global_numpy(loc_w, cam_w), cam_w)
    mid_global = to_global_numpy(relative_global_numpy(mid_local, cam_w), cam_w)
    if device_metrics:
        final_optimized_d = opt.engine.merge_windows(opt_global_d, 1, overlap=overlap, smooth=final_smooth is True)
        final_optimized_seq = final_optimized_d.cpu().numpy()
    else:
        final_optimized_seq = merge_batches(opt_global, overlap)
    final_estimated_seq = merge_batches(est_global, overlap)
    mid_local_pose_seq = merge_batches(mid_local, overlap)
    mid_estimated_seq = merge_batches(mid_global, overlap)
    final_gt_seq = merge_batches(cut_windows(gt, starts, seq_len), overlap)
    if final_smooth is True and not device_metrics:
        from .sequence import final_smooth as _smooth
        final_optimized_seq = _smooth(final_optimized_seq)
    if save_pose:
        # optimizer.py:469-483: out/<dataset>/<sequence>/result_pose.pkl under the working directory, the four sequences in the
        # containers the reference pickles (merge_batches' lists of [15,3] frames; the optimised one an ndarray after the smoothing)
        dataset_dir, seq_name = os.path.split(data_id)
        out_dir = "out/{}/{}".format(os.path.split(dataset_dir)[1], seq_name)
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "result_pose.pkl"), "wb") as f:
            pickle.dump({"estimated_pose": list(final_estimated_seq),
                         "optimized_pose": final_optimized_seq if final_smooth is True else list(np.asarray(final_optimized_seq)),
                         "mid_optimized_pose": list(mid_estimated_seq), "gt_pose": list(final_gt_seq)}, f)
    if device_metrics:
        errors = opt.engine.calculate_errors(final_estimated_seq, mid_estimated_seq, final_optimized_d, final_gt_seq)
    else:
        errors = calculate_errors(final_estimated_seq, mid_estimated_seq, final_optimized_seq, final_gt_seq)
    # The reference's merge_batches returns a LIST of [15,3] frames (optimizer.py:425-437); only the final Gaussian smoothing
    # (optimizer.py:448-450: gaussian_filter1d) turns final_optimized_seq into an ndarray -- same container types here.
    opt_out = final_optimized_seq if final_smooth is True else list(np.asarray(final_optimized_seq))
    res = (errors, list(final_estimated_seq), list(mid_local_pose_seq), opt_out, list(final_gt_seq))
    return res + (stats,) if return_stats else res
