"""CPU oracle for the GlobalEgoMocap window optimiser -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain-numpy restatement (float32 where the reference computes in float32) of the
reference's per-window hot path.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import this package; the shipped path
(`globalegomocap_amd/`) never does and fails loudly when its HIP library is missing.

Parity pin: every function below is checked against golden vectors produced by importing the
unmodified reference in the build container (`oracle/make_golden.py` -> `tests/golden/*.npz`,
checked by `tests/test_oracle_golden.py`).

Reference map (all paths relative to the reference tree):
  fold_vae / encode / decode        networks/models/SeqConvVAE.py:11-140 (+ torch BatchNorm1d eval,
                                    LeakyReLU(0.01), Conv1d / ConvTranspose1d k=3 s=1 p=1)
  fisheye_project                   utils/fisheye/FishEyeCalibrated.py:96-129
  bilinear_sample                   optimizer.py:139-149 (torch grid_sample, bilinear, zeros,
                                    align_corners=True)
  energy_and_grad                   optimizer.py:172-177,202-213,226-240
  lbfgs_strong_wolfe                torch/optim/lbfgs.py (third-party, torch 2.10: _cubic_interpolate,
                                    _strong_wolfe, LBFGS.step) as called at optimizer.py:261-270
  optimize_stage                    optimizer.py:242-276
  relative_global / to_global       utils/utils.py:62-66,99-112; optimizer.py:302-308
  optimize_sequence                 optimizer.py:311-450 (window loop, merge_batches, final smooth)
  lift_skeleton                     utils/skeleton.py:32-45,74-90,176-204; utils/fisheye/FishEyeCalibrated.py:18-33
                                    (pinned by oracle/make_golden_lift.py -> tests/golden/lift.npz)
"""
from dataclasses import dataclass, field

import numpy as np

PARENTS = np.array([0, 0, 1, 2, 0, 4, 5, 1, 7, 8, 9, 4, 11, 12, 13])
BN_EPS = 1e-5
SLOPE = np.float32(0.01)
F32 = np.float32


# --------------------------------------------------------------------------------------
# VAE (SeqConvVAE.py)
# --------------------------------------------------------------------------------------
def _a(t):
    if hasattr(t, "detach"):
        t = t.detach().cpu().numpy()
    return np.asarray(t)


def _fold_bn(w_taps, b, sd, prefix):
    """conv followed by eval-mode BatchNorm1d -> one affine conv.  w_taps: [3][Cin][Cout] f64."""
    g = _a(sd[prefix + ".weight"]).astype(np.float64)
    beta = _a(sd[prefix + ".bias"]).astype(np.float64)
    mu = _a(sd[prefix + ".running_mean"]).astype(np.float64)
    var = _a(sd[prefix + ".running_var"]).astype(np.float64)
    s = g / np.sqrt(var + BN_EPS)
    return w_taps * s[None, None, :], (b - mu) * s + beta


def _conv_taps(w):
    """Conv1d weight [Cout,Cin,3] -> taps[k][Cin][Cout] with out[t] = sum_k in[t+k-1] @ taps[k]."""
    return np.transpose(w.astype(np.float64), (2, 1, 0))


def _convT_taps(w):
    """ConvTranspose1d (s=1,p=1) weight [Cin,Cout,3]: out[t] = sum_k in[t+1-k] @ w[:,:,k]
    == conv with taps[k'] = w[:,:,2-k'] (SeqConvVAE.py:70-75)."""
    return np.transpose(w.astype(np.float64), (2, 0, 1))[::-1]


@dataclass
class FoldedVAE:
    T: int
    D: int
    enc: list            # [(taps f32 [3][Cin][Cout], bias f32 [Cout])]   all followed by LeakyReLU
    fc_mu: tuple         # (W f32 [D, C*T] indexed c*T+t, b)
    fc_var: tuple
    dec_in: tuple        # (W f32 [C*T, D], b)
    dec: list            # conv layers of the decoder; the last one has no activation
    hidden: tuple = field(default=())


def fold_vae(sd, seq_len=10):
    hidden, i = [], 0
    while ("encoder.%d.0.weight" % i) in sd:
        hidden.append(_a(sd["encoder.%d.0.weight" % i]).shape[0])
        i += 1
    enc = []
    for i in range(len(hidden)):
        w, b = _fold_bn(_conv_taps(_a(sd["encoder.%d.0.weight" % i])),
                        _a(sd["encoder.%d.0.bias" % i]).astype(np.float64), sd, "encoder.%d.1" % i)
        enc.append((w.astype(F32), b.astype(F32)))
    dec = []
    for i in range(len(hidden) - 1):
        w, b = _fold_bn(_convT_taps(_a(sd["decoder.%d.0.weight" % i])),
                        _a(sd["decoder.%d.0.bias" % i]).astype(np.float64), sd, "decoder.%d.1" % i)
        dec.append((w.astype(F32), b.astype(F32)))
    w, b = _fold_bn(_convT_taps(_a(sd["final_layer.0.weight"])),
                    _a(sd["final_layer.0.bias"]).astype(np.float64), sd, "final_layer.1")
    dec.append((w.astype(F32), b.astype(F32)))
    dec.append((_conv_taps(_a(sd["final_layer.3.weight"])).astype(F32), _a(sd["final_layer.3.bias"]).astype(F32)))
    f = lambda n: (_a(sd[n + ".weight"]).astype(F32), _a(sd[n + ".bias"]).astype(F32))
    D = _a(sd["fc_mu.weight"]).shape[0]
    return FoldedVAE(T=seq_len, D=D, enc=enc, fc_mu=f("fc_mu"), fc_var=f("fc_var"),
                     dec_in=f("decoder_input"), dec=dec, hidden=tuple(hidden))


def _conv3(x, taps, bias):
    """x [B,T,Cin] f32 -> [B,T,Cout]; zero padding in time."""
    B, T, _ = x.shape
    y = np.broadcast_to(bias, (B, T, taps.shape[2])).astype(F32).copy()
    y += x @ taps[1]
    y[:, 1:] += x[:, :-1] @ taps[0]
    y[:, :-1] += x[:, 1:] @ taps[2]
    return y


def _conv3_back(dy, taps):
    """adjoint of _conv3 w.r.t. x."""
    dx = dy @ taps[1].T
    dx[:, :-1] += dy[:, 1:] @ taps[0].T
    dx[:, 1:] += dy[:, :-1] @ taps[2].T
    return dx


def _lrelu(x):
    return np.where(x > 0, x, x * SLOPE).astype(F32)


def encode(vae, pose):
    """pose [B,T,45] f32 -> (mu, logvar) [B,D]  (SeqConvVAE.py:97-116,184-186)."""
    h = pose.astype(F32)
    for taps, b in vae.enc:
        h = _lrelu(_conv3(h, taps, b))
    flat = np.transpose(h, (0, 2, 1)).reshape(h.shape[0], -1)       # channel-major: c*T+t
    mu = flat @ vae.fc_mu[0].T + vae.fc_mu[1]
    logvar = flat @ vae.fc_var[0].T + vae.fc_var[1]
    return mu.astype(F32), logvar.astype(F32)


def latent_from_pose(vae, pose, eps):
    """z0 = mu + eps * exp(0.5 logvar)  (SeqConvVAE.py:159-169,184-189; eps is explicit here)."""
    mu, logvar = encode(vae, pose)
    return (eps.astype(F32) * np.exp(F32(0.5) * logvar) + mu).astype(F32)


def decode(vae, z, keep=False):
    """z [B,D] -> X [B,T,15,3] f32 (SeqConvVAE.py:131-140). keep=True also returns activations."""
    B = z.shape[0]
    h0 = (z.astype(F32) @ vae.dec_in[0].T + vae.dec_in[1]).astype(F32)
    h = np.transpose(h0.reshape(B, -1, vae.T), (0, 2, 1))            # view [B,C,T] -> [B,T,C]
    acts = []
    for taps, b in vae.dec[:-1]:
        h = _lrelu(_conv3(np.ascontiguousarray(h), taps, b))
        acts.append(h)
    out = _conv3(h, *vae.dec[-1])
    X = out.reshape(B, vae.T, 15, 3)
    return (X, acts) if keep else X


def decode_backward(vae, dX, acts):
    """dL/dX [B,T,15,3] -> dL/dz [B,D] (backward-data only; the VAE is frozen)."""
    B = dX.shape[0]
    g = _conv3_back(dX.reshape(B, vae.T, -1).astype(F32), vae.dec[-1][0])
    for (taps, _), a in zip(reversed(vae.dec[:-1]), reversed(acts)):
        g = g * np.where(a > 0, F32(1), SLOPE)
        g = _conv3_back(g, taps)
    g0 = np.transpose(g, (0, 2, 1)).reshape(B, -1)                   # back to c*T+t
    return (g0 @ vae.dec_in[0]).astype(F32)


# --------------------------------------------------------------------------------------
# Energy terms (optimizer.py:139-240) and their analytic gradient
# --------------------------------------------------------------------------------------
@dataclass
class Camera:
    poly: np.ndarray     # polynomialW2C
    cx: float
    cy: float


@dataclass
class Weights:
    w3d: float
    smooth: float
    bone: float
    vae: float
    reproj: float


LOCAL_W = Weights(w3d=0.01 / 10000, smooth=0.001 / 100, bone=0.01, vae=0.0, reproj=0.01)   # optimizer.py:355-358
GLOBAL_W = Weights(w3d=0.01, smooth=0.001, bone=0.01, vae=0.0, reproj=0.0)                  # optimizer.py:352-353


def fisheye_project(cam, P, want_jac=False):
    """P [n,3] f32 -> uv [n,2] f32 (FishEyeCalibrated.py:96-129); optional d(uv)/dP [n,2,3]."""
    x, y, z = P[:, 0].astype(F32), P[:, 1].astype(F32), P[:, 2].astype(F32)
    zz = -z
    n = np.sqrt(x * x + y * y).astype(F32)
    if not (n != 0).all():
        raise Exception("norm is zero!")
    theta = np.arctan(zz / n).astype(F32)
    inv = (F32(1) / n).astype(F32)
    rho = np.full_like(theta, F32(cam.poly[0]))
    drho = np.zeros_like(theta)
    t_i = np.ones_like(theta)
    for i in range(1, len(cam.poly)):
        drho = (drho + F32(i * cam.poly[i]) * t_i).astype(F32)
        t_i = (t_i * theta).astype(F32)
        rho = (rho + t_i * F32(cam.poly[i])).astype(F32)
    u = x * inv * rho + F32(cam.cx)
    v = y * inv * rho + F32(cam.cy)
    uv = np.stack([u, v], axis=1).astype(F32)
    if not want_jac:
        return uv
    r2 = n * n + zz * zz
    dth_dn = -zz / r2
    dth_dz = -n / r2                       # d theta / d z   (zz = -z)
    cx_, cy_ = x * inv, y * inv            # unit direction in the image plane
    # d(c*rho)/dx = rho * dc/dx + c * drho * dtheta/dn * dn/dx
    J = np.empty((P.shape[0], 2, 3), dtype=F32)
    J[:, 0, 0] = rho * (inv - x * x * inv ** 3) + cx_ * drho * dth_dn * cx_
    J[:, 0, 1] = rho * (-x * y * inv ** 3) + cx_ * drho * dth_dn * cy_
    J[:, 0, 2] = cx_ * drho * dth_dz
    J[:, 1, 0] = rho * (-x * y * inv ** 3) + cy_ * drho * dth_dn * cx_
    J[:, 1, 1] = rho * (inv - y * y * inv ** 3) + cy_ * drho * dth_dn * cy_
    J[:, 1, 2] = cy_ * drho * dth_dz
    return uv, J


def bilinear_sample(heat, ix, iy):
    """heat [n,H,W] f32, pixel coords -> (value, d/dix, d/diy); zeros outside (grid_sample)."""
    n, H, W = heat.shape
    x0 = np.floor(ix).astype(np.int64)
    y0 = np.floor(iy).astype(np.int64)
    fx = (ix - x0).astype(F32)
    fy = (iy - y0).astype(F32)
    idx = np.arange(n)

    def tex(yy, xx):
        ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
        return np.where(ok, heat[idx, np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)], F32(0)).astype(F32)

    nw, ne, sw, se = tex(y0, x0), tex(y0, x0 + 1), tex(y0 + 1, x0), tex(y0 + 1, x0 + 1)
    gx, gy = F32(1) - fx, F32(1) - fy
    val = nw * gx * gy + ne * fx * gy + sw * gx * fy + se * fx * fy
    dix = (ne - nw) * gy + (se - sw) * fy
    diy = (sw - nw) * gx + (se - ne) * fx
    return val.astype(F32), dix.astype(F32), diy.astype(F32)


def heat_coords(uv, H, W):
    """image px -> heat-map px exactly as optimizer.py:143-147 + grid_sample's unnormalise."""
    gx = ((uv[:, 0] - F32(128)) - F32(512)) / F32(512)
    gy = (uv[:, 1] - F32(512)) / F32(512)
    ix = ((gx + F32(1)) / F32(2)) * F32(W - 1)
    iy = ((gy + F32(1)) / F32(2)) * F32(H - 1)
    return ix.astype(F32), iy.astype(F32)


def mean_bone_length(skel):
    """skel [N,15,3] f32 -> [15] f32 (optimizer.py:42-43,89-94)."""
    s = skel.astype(F32).reshape(-1, 15, 3)
    return np.linalg.norm(s - s[:, PARENTS], axis=-1).astype(F32).mean(axis=0, dtype=F32)


def energy_and_grad(X, X_init, mean_bone, w, cam=None, heat=None):
    """One window.  X, X_init [T,15,3] f32; heat [T,H,W,15] (pickle layout) or None.

    Returns (total f64, parts f64[5] = (E_3d, E_smooth, E_bone, E_vae, E_reproj), dL/dX f32).
    Energies are accumulated in float64 from float32 terms (the reference sums in float32 and
    casts the total to a python float, lbfgs.py `float(closure())`).
    """
    X = X.astype(F32)
    T = X.shape[0]
    d3 = X - X_init.astype(F32)
    e3d = float(np.sum(np.square(d3), dtype=np.float64))
    g = F32(2 * w.w3d) * d3

    acc = X[:-2] - F32(2) * X[1:-1] + X[2:]                      # optimizer.py:202-208
    esm = float(np.sum(np.square(acc), dtype=np.float64))
    ws2 = F32(2 * w.smooth)
    g[:-2] += ws2 * acc
    g[1:-1] -= F32(2) * ws2 * acc
    g[2:] += ws2 * acc

    bone = X - X[:, PARENTS]                                        # optimizer.py:172-177
    ln = np.sqrt(np.sum(bone * bone, axis=-1)).astype(F32)
    diff = ln - mean_bone.astype(F32)[None]
    ebone = float(np.sum(np.square(diff), dtype=np.float64))
    safe = np.where(ln > 0, ln, F32(1))
    coef = np.where(ln > 0, F32(2 * w.bone) * diff / safe, F32(0)).astype(F32)   # norm' at 0 := 0
    gb = coef[..., None] * bone
    g += gb
    np.subtract.at(g, (slice(None), PARENTS), gb)

    evae = float(np.sum(np.square(X), dtype=np.float64))           # optimizer.py:238 (on the pose)
    if w.vae != 0:
        g += F32(2 * w.vae) * X

    erep = 0.0
    if w.reproj != 0:
        H, W = heat.shape[1], heat.shape[2]
        uv, J = fisheye_project(cam, X.reshape(-1, 3), want_jac=True)
        ix, iy = heat_coords(uv, H, W)
        hm = np.transpose(heat, (0, 3, 1, 2)).reshape(T * 15, H, W).astype(F32)   # optimizer.py:251-252
        val, dix, diy = bilinear_sample(hm, ix, iy)
        erep = -float(np.sum(val, dtype=np.float64))
        sx = F32((W - 1) / 1024.0)
        sy = F32((H - 1) / 1024.0)
        gu = (-F32(w.reproj)) * dix * sx
        gv = (-F32(w.reproj)) * diy * sy
        g += (gu[:, None] * J[:, 0, :] + gv[:, None] * J[:, 1, :]).reshape(T, 15, 3)
    parts = np.array([e3d, esm, ebone, evae, erep])
    total = w.w3d * e3d + w.smooth * esm + w.bone * ebone + w.vae * evae + w.reproj * erep
    return total, parts, g.astype(F32)


# --------------------------------------------------------------------------------------
# L-BFGS with strong-Wolfe line search, restated as an evaluation-driven state machine
# (torch/optim/lbfgs.py).  One `advance()` call consumes one (loss, grad) evaluation and either
# finishes or emits the next trial point -- the same structure the HIP kernel uses, so that a
# whole batch of windows can be advanced in lock-step "evaluation rounds".
# --------------------------------------------------------------------------------------
@dataclass
class LBFGSOptions:
    lr: float = 2.0
    max_iter: int = 25
    max_eval: int = 31           # max_iter * 5 // 4
    tol_grad: float = 1e-7
    tol_change: float = 1e-6     # optimizer.py:253,261-262
    history: int = 100
    c1: float = 1e-4
    c2: float = 0.9
    ls_tol_change: float = 1e-9  # _strong_wolfe default; LBFGS.step does not forward tolerance_change


def cubic_interpolate(x1, f1, g1, x2, f2, g2, bounds=None):
    if bounds is not None:
        lo, hi = bounds
    else:
        lo, hi = (x1, x2) if x1 <= x2 else (x2, x1)
    d1 = g1 + g2 - 3 * (f1 - f2) / (x1 - x2)
    d2s = d1 * d1 - g1 * g2
    if d2s >= 0:
        d2 = np.sqrt(d2s)
        if x1 <= x2:
            m = x2 - (x2 - x1) * ((g2 + d2 - d1) / (g2 - g1 + 2 * d2))
        else:
            m = x1 - (x1 - x2) * ((g1 + d2 - d1) / (g1 - g2 + 2 * d2))
        return min(max(m, lo), hi)
    return (lo + hi) / 2.0


def _dot(a, b):
    return float(np.dot(a.astype(F32), b.astype(F32)))


class LBFGSMachine:
    """Per-window resumable L-BFGS.  Scalars are python floats (float64), vectors float32."""
    INIT, BRACKET, ZOOM, DONE = 0, 1, 2, 3

    def __init__(self, x0, opt=None):
        self.o = opt or LBFGSOptions()
        self.x = x0.astype(F32).copy()
        self.phase = self.INIT
        self.n_iter = 0
        self.evals = 0
        self.S, self.Y, self.ro = [], [], []
        self.H_diag = 1.0
        self.trial = self.x.copy()
        self.trace = []

    # -- helpers ----------------------------------------------------------------------
    def _emit(self, t):
        self.t = float(t)
        self.trial = (self.x + F32(self.t) * self.d).astype(F32)
        self.ls_evals += 1

    def _finish(self):
        self.phase = self.DONE
        self.trial = self.x.copy()

    def _start_iteration(self):
        """Top of the `while n_iter < max_iter` loop: direction, initial step, line-search start."""
        o = self.o
        self.n_iter += 1
        g = self.g
        if self.n_iter == 1:
            self.d = (-g).astype(F32)
            self.H_diag = 1.0
        else:
            y = (g - self.prev_g).astype(F32)
            s = (self.d * F32(self.t)).astype(F32)
            ys = _dot(y, s)
            if ys > 1e-10:
                if len(self.S) == o.history:
                    self.S.pop(0); self.Y.pop(0); self.ro.pop(0)
                self.Y.append(y); self.S.append(s); self.ro.append(1.0 / ys)
                self.H_diag = ys / _dot(y, y)
            k = len(self.S)
            al = [0.0] * k
            q = (-g).astype(F32)
            for i in range(k - 1, -1, -1):
                al[i] = _dot(self.S[i], q) * self.ro[i]
                q = (q - F32(al[i]) * self.Y[i]).astype(F32)
            r = (q * F32(self.H_diag)).astype(F32)
            for i in range(k):
                be = _dot(self.Y[i], r) * self.ro[i]
                r = (r + F32(al[i] - be) * self.S[i]).astype(F32)
            self.d = r
        self.prev_g = g.copy()
        self.prev_loss = self.loss
        if self.n_iter == 1:
            t = min(1.0, 1.0 / float(np.sum(np.abs(g), dtype=F32))) * o.lr
        else:
            t = o.lr
        self.gtd = _dot(g, self.d)
        if self.gtd > -o.tol_change:
            return self._finish()
        # _strong_wolfe prologue
        self.d_norm = float(np.max(np.abs(self.d)))
        self.max_ls = o.max_eval - self.evals
        self.ls_iter = 0
        self.ls_evals = 0
        self.t_prev, self.f_prev, self.g_prev, self.gtd_prev = 0.0, self.loss, g.copy(), self.gtd
        self.phase = self.BRACKET
        self.first_bracket_eval = True
        self._emit(t)

    def _end_line_search(self):
        """`t, f_new, g_new = bracket[low]` then the tail of LBFGS.step's loop body."""
        o = self.o
        lo = self.low
        t, self.loss, self.g = self.br_t[lo], self.br_f[lo], self.br_g[lo].copy()
        self.t = t
        self.x = (self.x + F32(t) * self.d).astype(F32)
        self.evals += self.ls_evals
        opt_cond = float(np.max(np.abs(self.g))) <= o.tol_grad
        self.trace.append((self.n_iter, self.evals, t, self.loss, float(np.max(np.abs(self.g)))))
        if (self.n_iter == o.max_iter or self.evals >= o.max_eval or opt_cond
                or float(np.max(np.abs(self.d * F32(t)))) <= o.tol_change
                or abs(self.loss - self.prev_loss) < o.tol_change):
            return self._finish()
        self._start_iteration()

    def _zoom_entry(self, done):
        self.ls_done = done
        self.insuf = False
        self.low, self.high = (0, 1) if self.br_f[0] <= self.br_f[-1] else (1, 0)
        self._zoom_head()

    def _zoom_head(self):
        o = self.o
        if self.ls_done or self.ls_iter >= self.max_ls:
            return self._end_line_search()
        bt = self.br_t
        if abs(bt[1] - bt[0]) * self.d_norm < o.ls_tol_change:
            return self._end_line_search()
        t = cubic_interpolate(bt[0], self.br_f[0], self.br_gtd[0], bt[1], self.br_f[1], self.br_gtd[1])
        hi, lo = max(bt), min(bt)
        eps = 0.1 * (hi - lo)
        if min(hi - t, t - lo) < eps:
            if self.insuf or t >= hi or t <= lo:
                t = hi - eps if abs(t - hi) < abs(t - lo) else lo + eps
                self.insuf = False
            else:
                self.insuf = True
        else:
            self.insuf = False
        self.phase = self.ZOOM
        self._emit(t)

    # -- the one entry point ---------------------------------------------------------------
    def advance(self, f_new, g_new):
        """Consume the evaluation of `self.trial`; afterwards `self.trial` is the next point to
        evaluate, or `self.phase == DONE` and `self.x` is the result."""
        o = self.o
        f_new = float(f_new)
        g_new = g_new.astype(F32)
        if self.phase == self.DONE:
            return
        if self.phase == self.INIT:
            self.loss, self.g = f_new, g_new.copy()
            self.evals = 1
            if float(np.max(np.abs(g_new))) <= o.tol_grad:
                return self._finish()
            return self._start_iteration()
        gtd_new = _dot(g_new, self.d)
        t = self.t
        if self.phase == self.BRACKET:
            if not self.first_bracket_eval:
                self.ls_iter += 1
            self.first_bracket_eval = False
            if self.ls_iter < self.max_ls:
                armijo_fail = f_new > (self.loss + o.c1 * t * self.gtd) or (self.ls_iter > 1 and f_new >= self.f_prev)
                if armijo_fail or (not abs(gtd_new) <= -o.c2 * self.gtd and gtd_new >= 0):
                    self.br_t = [self.t_prev, t]
                    self.br_f = [self.f_prev, f_new]
                    self.br_g = [self.g_prev, g_new.copy()]
                    self.br_gtd = [self.gtd_prev, gtd_new]
                    return self._zoom_entry(False)
                if abs(gtd_new) <= -o.c2 * self.gtd:
                    self.br_t, self.br_f, self.br_g, self.br_gtd = [t], [f_new], [g_new.copy()], [gtd_new]
                    return self._zoom_entry(True)
                lo_b = t + 0.01 * (t - self.t_prev)
                hi_b = t * 10
                t_next = cubic_interpolate(self.t_prev, self.f_prev, self.gtd_prev, t, f_new, gtd_new,
                                           bounds=(lo_b, hi_b))
                self.t_prev, self.f_prev, self.g_prev, self.gtd_prev = t, f_new, g_new.copy(), gtd_new
                return self._emit(t_next)
            # ran out of line-search iterations while bracketing
            self.br_t, self.br_f, self.br_g = [0.0, t], [self.loss, f_new], [self.prev_g.copy(), g_new.copy()]
            self.br_gtd = [self.gtd, gtd_new]
            return self._zoom_entry(False)
        # ZOOM
        self.ls_iter += 1
        lo, hi = self.low, self.high
        if f_new > (self.loss + o.c1 * t * self.gtd) or f_new >= self.br_f[lo]:
            self.br_t[hi], self.br_f[hi], self.br_g[hi], self.br_gtd[hi] = t, f_new, g_new.copy(), gtd_new
            self.low, self.high = (0, 1) if self.br_f[0] <= self.br_f[1] else (1, 0)
        else:
            if abs(gtd_new) <= -o.c2 * self.gtd:
                self.ls_done = True
            elif gtd_new * (self.br_t[hi] - self.br_t[lo]) >= 0:
                self.br_t[hi], self.br_f[hi], self.br_g[hi], self.br_gtd[hi] = \
                    self.br_t[lo], self.br_f[lo], self.br_g[lo], self.br_gtd[lo]
            self.br_t[lo], self.br_f[lo], self.br_g[lo], self.br_gtd[lo] = t, f_new, g_new.copy(), gtd_new
        self._zoom_head()


def lbfgs_strong_wolfe(fun, x0, opt=None):
    """Minimise fun(x)->(f,g) from x0 exactly like `LBFGS(...).step(closure)` (optimizer.py:261-270)."""
    m = LBFGSMachine(x0, opt)
    while m.phase != m.DONE:
        f, g = fun(m.trial)
        m.advance(f, g)
    return m.x, {"n_iter": m.n_iter, "func_evals": m.evals, "loss": getattr(m, "loss", None), "trace": m.trace}


# --------------------------------------------------------------------------------------
# Window stage and sequence pipeline (optimizer.py:242-276, 311-450)
# --------------------------------------------------------------------------------------
def optimize_stage(vae, cam, w, pose, heat, mean_bone, eps, opt=None):
    """One `optimize_pose_seq_pytorch_LBFGS` call.  pose [T,15,3]; heat [T,H,W,15]; eps [D]."""
    X0 = pose.astype(F32)
    z0 = latent_from_pose(vae, X0.reshape(1, X0.shape[0], 45), eps.reshape(1, -1))[0]

    def fun(z):
        X, acts = decode(vae, z[None], keep=True)
        f, _, dX = energy_and_grad(X[0], X0, mean_bone, w, cam, heat)
        return f, decode_backward(vae, dX[None], acts)[0]

    z, stats = lbfgs_strong_wolfe(fun, z0, opt)
    return decode(vae, z[None])[0].astype(F32), stats


def transform_pose(pose, M):
    """utils/utils.py:62-66 (float64)."""
    return (np.concatenate([pose, np.ones((pose.shape[0], 1))], axis=1) @ M.T)[:, :3]


def relative_global(local, cams):
    """X_rel[t] = C0^-1 C_t X_loc[t]   (utils/utils.py:99-112)."""
    inv0 = np.linalg.inv(cams[0])
    return np.asarray([transform_pose(np.asarray(p, dtype=np.float64), inv0.dot(c)) for p, c in zip(local, cams)])


def to_global(rel, cams):
    """X_glob[t] = C0 X_rel[t]   (optimizer.py:302-308)."""
    return np.asarray([transform_pose(np.asarray(p, dtype=np.float64), cams[0]) for p in rel])


def merge_batches(seq, overlap=2):
    """optimizer.py:425-437."""
    seq = np.asarray(seq)
    if overlap == 0:
        return np.concatenate(seq)
    out = list(seq[0][:-overlap])
    for i in range(len(seq) - 1):
        out.extend((seq[i][-overlap:] + seq[i + 1][:overlap]) / 2)
        out.extend(seq[i + 1][overlap:-overlap])
    out.extend(seq[-1][-overlap:])
    return np.asarray(out)


def optimize_sequence(data, vae_local, vae_global, cam, eps, w_local=LOCAL_W, w_global=GLOBAL_W,
                      seq_len=10, overlap=2, final_smooth=False, opt=None):
    """`main()` of optimizer.py:311-450 without file I/O and metrics.

    data: the dict of test_data.pkl; eps [2*n_windows, D] in the reference's draw order (window i:
    local stage then global stage).  Returns dict of merged sequences.
    """
    from scipy.ndimage import gaussian_filter1d
    est = np.asarray(data["estimated_local_skeleton"])
    gt = np.asarray(data["gt_global_skeleton"])
    cams = np.asarray(data["camera_pose_list"])
    heat = np.asarray(data["heatmap_list"])
    mb = mean_bone_length(est.astype(F32))
    acc = {k: [] for k in ("est", "mid_local", "mid", "opt", "gt", "est_local")}
    stats = []
    for wi, i in enumerate(range(0, len(est) - seq_len + 1, seq_len - overlap)):
        loc, cs, hs = est[i:i + seq_len], cams[i:i + seq_len], heat[i:i + seq_len]
        res_local, st_a = optimize_stage(vae_local, cam, w_local, loc, hs, mb, eps[2 * wi], opt)
        est_rel = relative_global(loc, cs)
        opt_rel = relative_global(res_local, cs)
        res_global, st_b = optimize_stage(vae_global, cam, w_global, opt_rel, hs, mb, eps[2 * wi + 1], opt)
        acc["est_local"].append(loc)
        acc["mid_local"].append(res_local)
        acc["est"].append(to_global(est_rel, cs))
        acc["mid"].append(to_global(opt_rel, cs))
        acc["opt"].append(to_global(res_global.reshape(-1, 15, 3), cs))
        acc["gt"].append(gt[i:i + seq_len])
        stats.append((st_a, st_b))
    out = {k: merge_batches(v, overlap) for k, v in acc.items()}
    if final_smooth:
        out["opt"] = gaussian_filter1d(out["opt"], sigma=1, axis=0)
    out["stats"] = stats
    return out


# ---------------------------------------------------------------------------------------------------
# Input lifting: heat-maps + depths -> estimated_local_skeleton (literal restatement, big image and all)
def lift_skeleton(heat_hwj, depth, poly_c2w, cx, cy, size=1024, pad=128):
    """heat_hwj [H,W,J] (H == W), depth [J] -> [J,3] float64.

    set_skeleton_from_file (utils/skeleton.py:74-90): cv2.resize(heatmap, (size, size), INTER_NEAREST) -- for an
    integer ratio that is a plain repeat of every texel -- then np.pad(128) on the width, transpose to [J,H,W];
    get_max_preds (:176-204): row-major argmax, (x, y) = (idx % width, idx // width), zeroed where max <= 0;
    camera2world (FishEyeCalibrated.py:18-33): ray (x - cx, y - cy, -polyval(C2W, r)) normalised, times depth."""
    heat = np.asarray(heat_hwj)
    H, W, J = heat.shape
    assert H == W and size % H == 0
    big = np.repeat(np.repeat(heat, size // H, axis=0), size // W, axis=1)
    big = np.pad(big, ((0, 0), (pad, pad), (0, 0)), "constant", constant_values=0).transpose(2, 0, 1)
    width = big.shape[2]
    flat = big.reshape(J, -1)
    idx = np.argmax(flat, axis=1)
    maxv = np.amax(flat, axis=1)
    preds = np.stack([idx % width, np.floor(idx / width)], axis=1).astype(np.float32)
    preds *= np.greater(maxv, 0.0).astype(np.float32)[:, None]
    pc = preds.astype(np.float64) - np.array([cx, cy])
    x, y = pc[:, 0], pc[:, 1]
    z = np.polyval(np.asarray(poly_c2w, dtype=np.float64)[::-1], np.sqrt(x * x + y * y))
    p3 = np.array([x, y, -z])
    return (p3 / np.linalg.norm(p3, axis=0) * np.asarray(depth, dtype=np.float64)).T
