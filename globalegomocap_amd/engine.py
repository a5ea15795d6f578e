"""Thin Python object over the C ABI: owns one `gem_handle`, hands torch device pointers to it.

PyTorch is used for device memory and streams only; every number of the path is produced by the
HIP kernels behind `libgem_hip.so`.
"""
import ctypes as C

import numpy as np
import torch

from . import _capi
from .camera import FisheyeCamera, DEFAULT_CALIBRATION
from .skeleton import KINEMATIC_PARENTS, N_JOINTS
from .vae import VAEShape, flatten_state_dict

LOCAL_STAGE, GLOBAL_STAGE = _capi.STAGE_LOCAL, _capi.STAGE_GLOBAL


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def energy_weights(w3d, smooth, bone, vae, reproj):
    return _capi.GemEnergyWeights(float(w3d), float(smooth), float(bone), float(vae), float(reproj))


class WindowEngine:
    """B independent windows per call; one engine per device."""

    def __init__(self, shape=None, camera=None, max_windows=256, heat_size=(64, 64), device=None):
        self.lib = _capi.load_library()
        if not torch.cuda.is_available():
            raise _capi.GemError("no HIP device visible: the window optimiser has no CPU path")
        self.shape = shape or VAEShape()
        self.camera = camera or FisheyeCamera.from_json(DEFAULT_CALIBRATION)
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.max_windows = int(max_windows)
        self.heat_size = tuple(heat_size)
        cfg = _capi.GemConfig()
        cfg.seq_len, cfg.n_joints, cfg.latent_dim = self.shape.seq_len, N_JOINTS, self.shape.latent_dim
        cfg.n_hidden = len(self.shape.hidden)
        for i, v in enumerate(self.shape.hidden):
            cfg.hidden[i] = v
        cfg.heat_h, cfg.heat_w = self.heat_size
        cfg.n_poly = len(self.camera.poly_w2c)
        for i, v in enumerate(self.camera.poly_w2c):
            cfg.poly[i] = v
        cfg.cx, cfg.cy = self.camera.cx, self.camera.cy
        for i, v in enumerate(KINEMATIC_PARENTS):
            cfg.parents[i] = v
        cfg.max_windows, cfg.device = self.max_windows, self.device.index
        self._h = C.c_void_p()
        _capi.check(self.lib.gem_create(C.byref(cfg), C.byref(self._h)), self.lib)
        self.T, self.D = self.shape.seq_len, self.shape.latent_dim
        self.precision = "f32"
        self._graphs, self._gstream, self._bufs = False, None, {}
        self._pinned = {}          # graph mode: signature (input addresses) -> the caller's input tensors, kept alive (see _pin)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.gem_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_texel_cache(self, on=True):
        """Diagnostic switch of the reprojection term's texel-block cache (gem_set_texel_cache); results do not depend on it."""
        _capi.check(self.lib.gem_set_texel_cache(self._h, 1 if on else 0), self.lib)

    def enable_graphs(self, on=True):
        """hipGraph replay of whole `optimize_windows` / `optimize_stage` calls (gem_graph_enable): the first call with a
        given signature runs eagerly, the second is captured, later ones are ONE graph launch instead of ~700 kernel launches.
        With graphs on, the calls run on a stream owned by the engine (the legacy default stream cannot be captured; the
        caller's current stream waits for it) and their OUTPUT tensors are engine-owned buffers that the next call with the
        same batch size overwrites -- which is what keeps the signature (all pointers) stable.  Inputs must be the same
        tensors from call to call to get replays."""
        _capi.check(self.lib.gem_graph_enable(self._h, 1 if on else 0), self.lib)
        self._graphs = bool(on)
        if not on:
            self._pinned.clear()          # (gem_graph_enable(0) dropped the captured calls)
        if on and self._gstream is None:
            self._gstream = torch.cuda.Stream(device=self.device)

    MAX_PINNED = 4

    def drop_graphs(self):
        """Forget every captured call (gem_graph_enable 0 / 1: synchronises the device) and release the input tensors they pinned."""
        _capi.check(self.lib.gem_graph_enable(self._h, 0), self.lib)
        if self._graphs:
            _capi.check(self.lib.gem_graph_enable(self._h, 1), self.lib)
        self._pinned.clear()

    def _pin(self, *tensors):
        """A captured call holds the ADDRESSES of the caller's input tensors.  If the caller freed them, a later tensor could land on
        the same addresses and the old graph would be replayed on it -- on ROCm 7.2 that ended in a GPU memory fault even for
        equally sized buffers.  So the engine keeps the inputs of every signature it has seen alive (at most
        MAX_PINNED distinct input sets; one more drops all graphs and pins).  `drop_graphs()` releases them explicitly."""
        if not self._graphs:
            return
        key = tuple(t.data_ptr() for t in tensors if t is not None)
        if key not in self._pinned:
            if len(self._pinned) >= self.MAX_PINNED:
                self.drop_graphs()
            self._pinned[key] = tensors

    def graph_stats(self):
        c, r = C.c_int64(), C.c_int64()
        _capi.check(self.lib.gem_graph_stats(self._h, C.byref(c), C.byref(r)), self.lib)
        return {"captures": c.value, "replays": r.value}

    def _out(self, key, shape, dtype, zero=False):
        """Output buffer: fresh per call, or (graphs on) one persistent buffer per (entry point, role, shape)."""
        if not self._graphs:
            return (torch.zeros if zero else torch.empty)(shape, device=self.device, dtype=dtype)
        k = (key, tuple(shape), dtype)
        if k not in self._bufs:
            self._bufs[k] = torch.zeros(shape, device=self.device, dtype=dtype)
        return self._bufs[k]

    def _staged(self, key, t):
        """Copy of `t` in a persistent engine-owned buffer (graph mode: stable pointers for per-call temporaries)."""
        buf = self._out(key, tuple(t.shape), t.dtype)
        buf.copy_(t)
        return buf

    def _call(self, fn):
        """Run fn(stream_ptr) on the caller's current stream, or (graphs on) on the engine's stream, ordered after
        everything already enqueued on the current stream and before everything enqueued on it afterwards."""
        if not self._graphs:
            return fn(_stream())
        cur = torch.cuda.current_stream()
        self._gstream.wait_stream(cur)
        with torch.cuda.stream(self._gstream):
            r = fn(C.c_void_p(self._gstream.cuda_stream))
        cur.wait_stream(self._gstream)
        return r

    def set_lanes(self, min_windows):
        """optimize_windows calls of at least `min_windows` windows run as two half-batches on two streams, half a round apart
        (gem_set_lanes; 0 = never = the default).  Same results, bit for bit, for thresholds >= 4352."""
        _capi.check(self.lib.gem_set_lanes(self._h, int(min_windows)), self.lib)

    def set_precision(self, mode):
        """'f32' (default) | 'bf16x3' (split-bf16 MFMA, fp32-grade) | 'bf16' for the wide decoder/encoder products."""
        _capi.check(self.lib.gem_set_precision(self._h, _capi.PRECISION[mode]), self.lib)
        self.precision = mode

    # ------------------------------------------------------------------ weights
    def load_vae(self, stage, state_dict):
        blobs = flatten_state_dict(state_dict, self.shape)
        n = len(blobs)
        ptrs = (C.c_void_p * n)(*[b.ctypes.data_as(C.c_void_p) for b in blobs])
        sizes = (C.c_int64 * n)(*[b.size for b in blobs])
        _capi.check(self.lib.gem_load_vae(self._h, stage, n, ptrs, sizes), self.lib)

    # ------------------------------------------------------------------ helpers
    def _f32(self, a, shape=None):
        t = torch.as_tensor(np.asarray(a) if not torch.is_tensor(a) else a)
        t = t.to(device=self.device, dtype=torch.float32).contiguous()
        if shape is not None:
            t = t.reshape(shape)
        return t

    def _i32(self, a):
        return torch.as_tensor(np.asarray(a) if not torch.is_tensor(a) else a).to(device=self.device, dtype=torch.int32).contiguous()

    def _check_B(self, B):
        if B > self.max_windows:
            raise ValueError("B=%d exceeds max_windows=%d of this engine" % (B, self.max_windows))

    def _check_heat(self, heat_t, frame0, B):
        """Heat-map resolution and window starts against the frames actually held (the C ABI never learns n_frames: a wrong
        start would be a silent out-of-bounds read on the device).  frame0 is checked when it is host data (no sync)."""
        if heat_t is None:
            return
        if heat_t.dim() != 4 or tuple(heat_t.shape[1:]) != (self.heat_size[0], self.heat_size[1], N_JOINTS):
            raise ValueError("heat-maps must be [F,%d,%d,%d] (the engine's heat_size), got %s"
                             % (self.heat_size[0], self.heat_size[1], N_JOINTS, tuple(heat_t.shape)))
        if frame0 is None:
            raise ValueError("heat-maps need the first frame of every window (frame0)")
        if not torch.is_tensor(frame0) or not frame0.is_cuda:
            f = np.asarray(frame0.cpu() if torch.is_tensor(frame0) else frame0).reshape(-1)
            if f.shape[0] != B:
                raise ValueError("frame0 has %d entries for %d windows" % (f.shape[0], B))
            if B and (f.min() < 0 or f.max() + self.T > heat_t.shape[0]):
                raise ValueError("a window [frame0, frame0 + %d) leaves the %d frames of heat-maps" % (self.T, heat_t.shape[0]))

    def mean_bone_length(self, poses):
        p = self._f32(poses).reshape(-1, N_JOINTS, 3)
        out = torch.empty(N_JOINTS, device=self.device, dtype=torch.float32)
        _capi.check(self.lib.gem_mean_bone_length(self._h, _ptr(p), p.shape[0], _ptr(out), _stream()), self.lib)
        return out

    def encode(self, stage, pose, eps=None):
        pose = self._f32(pose).reshape(-1, self.T, N_JOINTS * 3)
        B = pose.shape[0]
        self._check_B(B)
        eps_t = self._f32(eps).reshape(B, self.D) if eps is not None else None
        mu, lv, z = (torch.empty(B, self.D, device=self.device) for _ in range(3))
        _capi.check(self.lib.gem_encode(self._h, stage, B, _ptr(pose), _ptr(eps_t), _ptr(mu), _ptr(lv), _ptr(z), _stream()),
                    self.lib)
        return mu, lv, z

    def decode(self, stage, z):
        z = self._f32(z).reshape(-1, self.D)
        B = z.shape[0]
        self._check_B(B)
        out = torch.empty(B, self.T, N_JOINTS, 3, device=self.device)
        _capi.check(self.lib.gem_decode(self._h, stage, B, _ptr(z), _ptr(out), _stream()), self.lib)
        return out

    def energy_grad(self, stage, z, pose_init, mean_bone, weights, heat=None, frame0=None):
        z = self._f32(z).reshape(-1, self.D)
        B = z.shape[0]
        self._check_B(B)
        p0 = self._f32(pose_init).reshape(B, self.T, N_JOINTS, 3)
        mb = self._f32(mean_bone).reshape(-1, N_JOINTS).expand(B, N_JOINTS).contiguous()
        heat_t = self._f32(heat) if heat is not None else None
        f0 = self._i32(frame0) if frame0 is not None else None
        self._check_heat(heat_t, frame0, B)
        E = torch.empty(B, device=self.device, dtype=torch.float64)
        parts = torch.empty(B, 5, device=self.device, dtype=torch.float64)
        dz = torch.empty(B, self.D, device=self.device)
        X = torch.empty(B, self.T, N_JOINTS, 3, device=self.device)
        _capi.check(self.lib.gem_energy_grad(self._h, stage, B, _ptr(z), _ptr(p0), _ptr(heat_t), _ptr(f0), _ptr(mb),
                                             C.byref(weights), _ptr(E), _ptr(parts), _ptr(dz), _ptr(X), _stream()), self.lib)
        return E, parts, dz, X

    def optimize_stage(self, stage, pose_in, mean_bone, eps, weights, heat=None, frame0=None, opts=None, want_stats=True):
        """One stage for B windows.  With graphs on, the small inputs (pose, mean bone, eps, frame0) are copied into
        engine-owned buffers so that the call's signature is stable from call to call (replay needs identical pointers);
        `heat` is passed through as it is and must be the same device tensor to get replays.  Outputs are kept per stage."""
        p = self._f32(pose_in).reshape(-1, self.T, N_JOINTS, 3)
        B = p.shape[0]
        self._check_B(B)
        mb = self._f32(mean_bone).reshape(-1, N_JOINTS).expand(B, N_JOINTS).contiguous()
        eps_t = self._f32(eps).reshape(B, self.D)
        heat_t = self._f32(heat) if heat is not None else None
        f0 = self._i32(frame0) if frame0 is not None else None
        self._check_heat(heat_t, frame0, B)
        if self._graphs:
            p, mb, eps_t = (self._staged("stage%d_%s" % (stage, k), t) for k, t in (("pose", p), ("mb", mb), ("eps", eps_t)))
            f0 = self._staged("stage%d_f0" % stage, f0) if f0 is not None else None
            self._pin(heat_t)
        opts = opts or _capi.default_lbfgs_opts()
        out = self._out("stage%d_out" % stage, (B, self.T, N_JOINTS, 3), torch.float32)
        stats = self._out("stage%d_stats" % stage, (B, 4), torch.int32, zero=True) if want_stats else None
        self._call(lambda st: _capi.check(self.lib.gem_optimize_stage(self._h, stage, B, _ptr(p), _ptr(heat_t), _ptr(f0), _ptr(mb),
                                                                      _ptr(eps_t), C.byref(weights), C.byref(opts), _ptr(out),
                                                                      _ptr(stats), st), self.lib))
        return out, stats

    def optimize_windows(self, local_pose, cams, heat, frame0, mean_bone, eps_local, eps_global, w_local, w_global,
                         opts=None, want_stats=True):
        """All tensors must already live on the device (this is the timed call of bench.py).

        local_pose [F,15,3] f32, cams [F,4,4] f64, heat [F,H,W,15] f32, frame0 [B] i32, mean_bone [B,15] f32,
        eps_* [B,D] f32 -> (mid_local [B,T,15,3] f32, global [B,T,15,3] f64, stats [2B,4] i32 or None)."""
        B = frame0.shape[0]
        self._check_B(B)
        for t, dt in ((local_pose, torch.float32), (cams, torch.float64), (heat, torch.float32), (frame0, torch.int32),
                      (mean_bone, torch.float32), (eps_local, torch.float32), (eps_global, torch.float32)):
            if t is not None and (t.dtype != dt or not t.is_cuda or not t.is_contiguous()):
                raise TypeError("optimize_windows wants contiguous device tensors of the documented dtypes")
        F = local_pose.shape[0]
        if cams.shape[0] != F or (heat is not None and heat.shape[0] != F):
            raise ValueError("optimize_windows: local_pose, cams and heat must cover the same frames (%d / %d / %s)"
                             % (F, cams.shape[0], None if heat is None else heat.shape[0]))
        if tuple(local_pose.shape[1:]) != (N_JOINTS, 3) or tuple(cams.shape[1:]) != (4, 4):
            raise ValueError("optimize_windows: local_pose must be [F,15,3] and cams [F,4,4]")
        if heat is not None and tuple(heat.shape[1:]) != (self.heat_size[0], self.heat_size[1], N_JOINTS):
            raise ValueError("optimize_windows: heat-maps must be [F,%d,%d,%d] (the engine's heat_size), got %s"
                             % (self.heat_size[0], self.heat_size[1], N_JOINTS, tuple(heat.shape)))
        if tuple(mean_bone.shape) != (B, N_JOINTS) or tuple(eps_local.shape) != (B, self.D) or tuple(eps_global.shape) != (B, self.D):
            raise ValueError("optimize_windows: mean_bone must be [B,15] and eps_* [B,%d] with B = %d windows" % (self.D, B))
        if F < self.T:
            raise ValueError("optimize_windows: %d frames cannot hold a %d-frame window" % (F, self.T))
        opts = opts or _capi.default_lbfgs_opts()
        self._pin(local_pose, cams, heat, frame0, mean_bone, eps_local, eps_global)
        mid = self._out("win_mid", (B, self.T, N_JOINTS, 3), torch.float32)
        glob = self._out("win_glob", (B, self.T, N_JOINTS, 3), torch.float64)
        stats = self._out("win_stats", (2 * B, 4), torch.int32, zero=True) if want_stats else None
        self._call(lambda st: _capi.check(self.lib.gem_optimize_windows(self._h, B, _ptr(local_pose), _ptr(cams), _ptr(heat),
                                                                        _ptr(frame0), _ptr(mean_bone), _ptr(eps_local),
                                                                        _ptr(eps_global), C.byref(w_local), C.byref(w_global),
                                                                        C.byref(opts), _ptr(mid), _ptr(glob), _ptr(stats), st),
                                          self.lib))
        return mid, glob, stats

    def read_trace(self, B, n_rounds=33):
        """Closure values of the last stage run on this engine: numpy [B, n_rounds] f64, NaN after a window's last
        evaluation (column r = evaluation round r).  For parity tests against the reference's closure traces."""
        self._check_B(B)
        out = torch.empty(n_rounds, B, device=self.device, dtype=torch.float64)
        _capi.check(self.lib.gem_read_trace(self._h, B, n_rounds, _ptr(out), _stream()), self.lib)
        return out.cpu().numpy().T.copy()

    # ------------------------------------------------------------------ sequence post-processing
    def _f64(self, a):
        t = torch.as_tensor(np.asarray(a) if not torch.is_tensor(a) else a)
        return t.to(device=self.device, dtype=torch.float64).contiguous()

    def merge_windows(self, windows, n_chunks, overlap=2, smooth=True):
        """merge_batches per chunk (+ gaussian_filter1d sigma=1 per chunk): [n_chunks*wpc,T,J,3] -> [n_chunks*fpc,J,3] f64."""
        w = self._f64(windows).reshape(-1, self.T, N_JOINTS, 3)
        if n_chunks < 1 or w.shape[0] % n_chunks:
            raise ValueError("merge_windows: %d windows do not split into %d chunks" % (w.shape[0], n_chunks))
        wpc = w.shape[0] // n_chunks
        fpc = wpc * (self.T - overlap) + overlap
        out = torch.empty(n_chunks * fpc, N_JOINTS, 3, device=self.device, dtype=torch.float64)
        _capi.check(self.lib.gem_merge_windows(self._h, _ptr(w), n_chunks, wpc, overlap, 1 if smooth else 0, _ptr(out), _stream()),
                    self.lib)
        return out

    ERROR_KEYS = ("original_global_mpjpe", "mid_global_mpjpe", "optimized_global_mpjpe", "original_camera_pos_error",
                  "optimized_camera_pos_error", "original_aligned_camera_pos_error", "mid_aligned_camera_pose_error",
                  "optimized_aligned_camera_pos_error", "original_aligned_global_mpjpe", "aligned_mid_seq_mpjpe",
                  "optimized_aligned_global_mpjpe", "aligned_original_mpjpe", "aligned_mid_optimized_mpjpe",
                  "aligned_optimized_mpjpe", "bone_length_aligned_original_mpjpe", "bone_length_aligned_mid_optimized_mpjpe",
                  "bone_length_aligned_optimized_mpjpe")

    def calculate_errors_device(self, est, mid, opt, gt):
        """calculate_errors on the device: returns the [17+J] f64 tensor (no synchronisation)."""
        from .skeleton import mean_bone_length_mm
        e, m, o, g = (self._f64(x).reshape(-1, N_JOINTS, 3) for x in (est, mid, opt, gt))
        if not (e.shape == m.shape == o.shape == g.shape):
            raise AssertionError("calculate_errors: sequences differ in shape")
        bone = np.ascontiguousarray(mean_bone_length_mm(), dtype=np.float64)
        out = torch.empty(17 + N_JOINTS, device=self.device, dtype=torch.float64)
        _capi.check(self.lib.gem_calculate_errors(self._h, _ptr(e), _ptr(m), _ptr(o), _ptr(g), e.shape[0],
                                                  bone.ctypes.data_as(C.POINTER(C.c_double)), _ptr(out), _stream()), self.lib)
        return out

    def calculate_errors_chunks(self, est, mid, opt, gt, n_chunks):
        """calculate_errors for each of `n_chunks` equally long sequences laid end to end (device f64 tensors [n_chunks*F,J,3]):
        [n_chunks, 17+J] f64 on the device, one library call, no synchronisation."""
        from .skeleton import mean_bone_length_mm
        for x in (est, mid, opt, gt):
            if not (torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float64 and x.is_contiguous() and x.shape == est.shape):
                raise TypeError("calculate_errors_chunks wants equally shaped contiguous float64 device tensors")
        total = est.numel() // (N_JOINTS * 3)
        if n_chunks < 1 or total % n_chunks:
            raise ValueError("calculate_errors_chunks: %d frames do not split into %d chunks" % (total, n_chunks))
        bone = np.ascontiguousarray(mean_bone_length_mm(), dtype=np.float64)
        out = torch.empty(n_chunks, 17 + N_JOINTS, device=self.device, dtype=torch.float64)
        _capi.check(self.lib.gem_calculate_errors_chunks(self._h, _ptr(est), _ptr(mid), _ptr(opt), _ptr(gt), n_chunks, total // n_chunks,
                                                         bone.ctypes.data_as(C.POINTER(C.c_double)), _ptr(out), _stream()), self.lib)
        return out

    def calculate_errors(self, est, mid, opt, gt):
        """Same keys and definitions as the reference's calculate_errors (calculate_errors.py:114-179)."""
        from collections import OrderedDict
        v = self.calculate_errors_device(est, mid, opt, gt).cpu().numpy()
        r = OrderedDict((k, float(v[i])) for i, k in enumerate(self.ERROR_KEYS))
        r["joints_error"] = v[17:].copy()
        return r

    # ------------------------------------------------------------------ input lifting
    def lift_skeleton(self, heat, depth, upscale=16, pad_x=128, pad_y=0, want_f64=True):
        """heat [F,H,W,15] f32 + depth [F,15] -> estimated_local_skeleton [F,15,3] (f64 like the reference's pickle,
        and f32 for optimize_windows): heat-map argmax + fisheye un-projection (utils/skeleton.py:32-45,176-204)."""
        heat_t = self._f32(heat)
        if heat_t.dim() != 4 or tuple(heat_t.shape[1:]) != (self.heat_size[0], self.heat_size[1], N_JOINTS):
            raise ValueError("lift_skeleton wants heat-maps [F,%d,%d,%d]" % (self.heat_size[0], self.heat_size[1], N_JOINTS))
        F = heat_t.shape[0]
        dep = self._f64(depth).reshape(F, N_JOINTS)
        poly = np.ascontiguousarray(self.camera.poly_c2w, dtype=np.float64)
        o64 = torch.empty(F, N_JOINTS, 3, device=self.device, dtype=torch.float64) if want_f64 else None
        o32 = torch.empty(F, N_JOINTS, 3, device=self.device, dtype=torch.float32)
        _capi.check(self.lib.gem_lift_skeleton(self._h, _ptr(heat_t), _ptr(dep), F, poly.ctypes.data_as(C.POINTER(C.c_double)),
                                               len(poly), upscale, pad_x, pad_y, _ptr(o64), _ptr(o32), _stream()), self.lib)
        return o64, o32

    # ------------------------------------------------------------------ profiling hook (bench.py)
    def profile_enable(self, on):
        _capi.check(self.lib.gem_profile_enable(self._h, 1 if on else 0), self.lib)

    def profile_kernels(self, family):
        """Names (as rocprofv3 prints them) of the kernels launched for `family` while profiling was on, since the last call."""
        buf = C.create_string_buffer(2048)
        _capi.check(self.lib.gem_profile_kernels(self._h, family, buf, len(buf)), self.lib)
        return buf.value.decode()

    def profile_read(self, family):
        ms, n, fl = C.c_double(), C.c_int64(), C.c_double()
        _capi.check(self.lib.gem_profile_read(self._h, family, C.byref(ms), C.byref(n), C.byref(fl)), self.lib)
        return ms.value, n.value, fl.value


def stats_to_numpy(stats):
    """[n,4] int32 tensor (n_iter, func_evals, final_loss bits, status) -> structured numpy.

    `status` is a bit field (gem_window_stats): bit 0 = the window's L-BFGS finished, bit 1 = a closure value was NaN (a joint
    on the optical axis: the reference raises "norm is zero!").  `finished` / `degenerate` are those two bits as booleans;
    `status == 1` therefore reads "finished and not degenerate"."""
    a = stats.cpu().numpy()
    out = np.zeros(a.shape[0], dtype=[("n_iter", "i4"), ("func_evals", "i4"), ("final_loss", "f4"), ("status", "i4"),
                                      ("finished", "?"), ("degenerate", "?")])
    out["n_iter"], out["func_evals"], out["status"] = a[:, 0], a[:, 1], a[:, 3]
    out["final_loss"] = a[:, 2].copy().view(np.float32)
    out["finished"], out["degenerate"] = (a[:, 3] & 1) != 0, (a[:, 3] & 2) != 0
    return out


def raise_if_degenerate(stats):
    """The reference's `Exception("norm is zero!")` (FishEyeCalibrated.py:124-127) for users of the engine's own calls: pass the
    stats tensor (or its stats_to_numpy) of optimize_stage / optimize_windows.  Synchronises on the stats."""
    st = stats if isinstance(stats, np.ndarray) else stats_to_numpy(stats)
    if st["degenerate"].any() or not np.isfinite(st["final_loss"]).all():
        raise Exception("norm is zero!")
