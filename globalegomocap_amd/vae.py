"""Motion-VAE checkpoint schema (host side).

The hot path consumes the reference's VAE checkpoints unchanged:
`torch.load(path)['state_dict']` (`optimizer.py:59-60`, written by `networks/train.py:102-108`)
with the key names of `ConvVAE` (`networks/models/SeqConvVAE.py:11-92`).  This module describes
that schema, turns a state_dict into the flat list of fp32 blobs the C ABI takes
(`gem_load_vae`, BatchNorm folding happens inside the library), and can synthesise
random weights in the same schema from a numpy seed (the trained checkpoints are an external
download and are not available offline).
"""
from collections import OrderedDict
from dataclasses import dataclass

import numpy as np

DEFAULT_HIDDEN = (64, 64, 128, 256, 512)   # SeqConvVAE.py:29-30
BN_EPS = 1e-5                              # torch.nn.BatchNorm1d default
LEAKY_SLOPE = 0.01                         # torch.nn.LeakyReLU default


@dataclass(frozen=True)
class VAEShape:
    channels: int = 45          # in_channels == out_channels (15 joints x 3)
    latent_dim: int = 2048
    seq_len: int = 10
    hidden: tuple = DEFAULT_HIDDEN

    @property
    def flat_dim(self):
        return self.hidden[-1] * self.seq_len

    def conv_layers(self):
        """(key prefix, kind, c_in, c_out, has_bn) in execution order, encoder then decoder."""
        enc, c = [], self.channels
        for i, h in enumerate(self.hidden):
            enc.append(("encoder.%d" % i, "conv", c, h, True))
            c = h
        rev = tuple(reversed(self.hidden))
        dec = []
        for i in range(len(rev) - 1):
            dec.append(("decoder.%d" % i, "convT", rev[i], rev[i + 1], True))
        dec.append(("final_layer", "convT", rev[-1], rev[-1], True))
        dec.append(("final_layer.3", "conv", rev[-1], self.channels, False))
        return enc, dec

    def schema(self):
        """OrderedDict name -> shape of every tensor of the state_dict (float tensors only)."""
        s = OrderedDict()

        def bn(prefix, c):
            s[prefix + ".weight"] = (c,)
            s[prefix + ".bias"] = (c,)
            s[prefix + ".running_mean"] = (c,)
            s[prefix + ".running_var"] = (c,)

        enc, dec = self.conv_layers()
        for prefix, _, ci, co, _ in enc:
            s[prefix + ".0.weight"] = (co, ci, 3)
            s[prefix + ".0.bias"] = (co,)
            bn(prefix + ".1", co)
        for name in ("fc_mu", "fc_var"):
            s[name + ".weight"] = (self.latent_dim, self.flat_dim)
            s[name + ".bias"] = (self.latent_dim,)
        s["decoder_input.weight"] = (self.flat_dim, self.latent_dim)
        s["decoder_input.bias"] = (self.flat_dim,)
        for prefix, kind, ci, co, has_bn in dec:
            if prefix == "final_layer.3":
                s[prefix + ".weight"] = (co, ci, 3)
                s[prefix + ".bias"] = (co,)
            else:
                s[prefix + ".0.weight"] = (ci, co, 3)     # ConvTranspose1d: [C_in, C_out, k]
                s[prefix + ".0.bias"] = (co,)
                bn(prefix + ".1", co)
        return s


def infer_shape(state_dict, seq_len=10):
    """Recover (latent_dim, hidden) from a state_dict in the reference schema."""
    hidden, i = [], 0
    while ("encoder.%d.0.weight" % i) in state_dict:
        hidden.append(int(_np(state_dict["encoder.%d.0.weight" % i]).shape[0]))
        i += 1
    if not hidden:
        raise KeyError("state_dict has no 'encoder.0.0.weight': not a ConvVAE checkpoint")
    channels = int(_np(state_dict["encoder.0.0.weight"]).shape[1])
    latent, flat = _np(state_dict["fc_mu.weight"]).shape
    if flat != hidden[-1] * seq_len:
        raise ValueError("fc_mu expects %d inputs, hidden[-1]*seq_len = %d" % (flat, hidden[-1] * seq_len))
    return VAEShape(channels=channels, latent_dim=int(latent), seq_len=seq_len, hidden=tuple(hidden))


def _np(t):
    if hasattr(t, "detach"):
        t = t.detach().cpu().numpy()
    return np.asarray(t)


def check_state_dict(state_dict, shape):
    """Raise like `load_state_dict` would on missing keys or shape mismatch (optimizer.py:60)."""
    missing, bad = [], []
    for name, shp in shape.schema().items():
        if name not in state_dict:
            missing.append(name)
        elif tuple(_np(state_dict[name]).shape) != tuple(shp):
            bad.append("%s: %s vs %s" % (name, tuple(_np(state_dict[name]).shape), tuple(shp)))
    if missing or bad:
        raise RuntimeError("Error(s) in loading state_dict for ConvVAE: missing %s; size mismatch %s"
                           % (missing, bad))


def flatten_state_dict(state_dict, shape):
    """state_dict -> list of contiguous fp32 arrays in `shape.schema()` order (the C-ABI blob order)."""
    check_state_dict(state_dict, shape)
    return [np.ascontiguousarray(_np(state_dict[name]), dtype=np.float32) for name in shape.schema()]


def synthetic_state_dict(shape, seed, gain=1.0):
    """Seeded random weights in the reference schema (numpy PCG64: identical on every machine).

    Linear/conv weights ~ U(+-gain/sqrt(fan_in)); BatchNorm statistics are non-trivial so that the
    folding is exercised.  Returns an OrderedDict of float32 numpy arrays (plus the integer
    `num_batches_tracked` entries torch expects).
    """
    rng = np.random.default_rng(seed)
    sd = OrderedDict()
    for name, shp in shape.schema().items():
        leaf = name.rsplit(".", 1)[1]
        if leaf == "weight" and len(shp) >= 2:
            if name.startswith(("decoder.", "final_layer.0")):
                fan_in = shp[0] * shp[2]            # ConvT weight is [C_in, C_out, k]
            else:
                fan_in = int(np.prod(shp[1:]))
            b = gain / np.sqrt(fan_in)
            a = rng.uniform(-b, b, size=shp)
        elif leaf == "weight":                       # BatchNorm gamma
            a = rng.uniform(0.8, 1.2, size=shp)
        elif leaf == "running_var":
            a = rng.uniform(0.5, 1.5, size=shp)
        elif leaf == "running_mean":
            a = rng.uniform(-0.1, 0.1, size=shp)
        else:                                        # biases (conv, linear, BN beta)
            a = rng.uniform(-0.05, 0.05, size=shp)
        sd[name] = a.astype(np.float32)
        if leaf == "running_var":
            sd[name.rsplit(".", 1)[0] + ".num_batches_tracked"] = np.asarray(1, dtype=np.int64)
    return sd


def structured_state_dict(shape, seed, latent_gain=8.0, sigma0=0.00125, coupling=0.01, spread=5.0, feature_offset=0.0,
                          mean_pose=None, pose_scale=0.3, signal_offset=3.0):
    """A full-size, WELL-CONDITIONED motion VAE built deterministically from a seed -- no training.

    Purpose: parity fixtures at the reference's real size (D = 2048).  A random-init VAE decodes far from its input
    and a fitted one cannot be regenerated bit-identically on another machine; this construction can: it uses only
    the PCG64 integer / uniform streams and IEEE +,-,*,/,sqrt on float64 scalars and arrays elementwise (no BLAS,
    LAPACK, reductions or transcendental functions), so `state_dict_sha256` of the result is the same everywhere.

    Structure (same graph and key names as ConvVAE, networks/models/SeqConvVAE.py:11-140):
      * signal path: channels 0..C-1 of every conv layer carry  OFF + u_j(t),  u = (x - mean_pose) / pose_scale,
        through the centre tap; the offset keeps them on the positive branch of LeakyReLU, so the path is affine.
        Each conv's weight / bias are solved through its (non-trivial, seeded) BatchNorm so that the folded layer
        realises exactly that map.
      * latent: mu = Q L^-1 u_flat / latent_gain with Q [D, T*C] = sign-flipped columns of the D x D Hadamard matrix /
        sqrt(D) (orthonormal columns, dense) and L = diag(lambda_i), lambda_i in (1/spread, 1] seeded; decoder_input =
        latent_gain * L Q^T: decode(encode(x)) = x on the signal path, and the decoder's singular values span a factor
        `spread` (a trained decoder is far from isotropic; an isotropic one lets L-BFGS finish in a handful of steps).
      * feature channels (the remaining ones): seeded three-tap weights on DIFFERENCES of channel pairs (2i, 2i+1) of
        the layer below -- a uniform offset cancels exactly -- feeding back into the signal channels and into mu with
        gain `coupling`.  feature_offset = 0: they straddle the LeakyReLU kink, the decoder is mildly non-linear like a
        fitted VAE (reconstruction to a few millimetres).  feature_offset > 0 (say 3): they stay on the positive branch
        too, the whole network is affine and the energy without reprojection term is SMOOTH: trajectories of two
        implementations then stay together to rounding, instead of parting at the first kink they cross on different sides.
      * logvar = log(sigma0^2) (+ small seeded weights): z0 = mu + eps * sigma0 moves the pose by ~latent_gain*sigma0*pose_scale.

    signal_offset (OFF, default 3: the gauge of the committed reference goldens): bf16 activations resolve a signal channel to
    2^-8 of its VALUE, i.e. of OFF + u -- with OFF = 3 and pose_scale = 0.3 m that is up to 4.7 mm of pose per rounding, an
    artefact of the gauge, not of the network.  OFF = 1 keeps the signal on the positive branch for |x - mean_pose| < 0.3 m (the
    synthetic motion stays within 0.2 m) and brings the rounding down to 0.6-1.2 mm: the "bf16-friendly" variant the bf16 parity
    tests use against the fp32 oracle.  Same function in exact arithmetic for any OFF.
    """
    from .skeleton import MEAN3D_MM
    C, T, D = shape.channels, shape.seq_len, shape.latent_dim
    n_sig = T * C
    if D & (D - 1) or D < n_sig or min(shape.hidden) < C + 2:
        raise ValueError("structured_state_dict needs latent_dim = 2^k >= seq_len*channels and hidden widths > channels")
    rng = np.random.default_rng(seed)
    if mean_pose is None:
        mean_pose = (MEAN3D_MM.T / 1000.0).reshape(-1)[:C]
    m = np.asarray(mean_pose, dtype=np.float64)
    OFF, FOFF, g = float(signal_offset), float(feature_offset), float(latent_gain)

    def uni(shp, lo, hi):                      # U[lo, hi) from the raw 53-bit uniform stream
        return lo + (hi - lo) * rng.random(size=shp)

    def bn_params(c):
        return dict(weight=uni((c,), 0.8, 1.2), bias=uni((c,), -0.05, 0.05), running_mean=uni((c,), -0.1, 0.1),
                    running_var=uni((c,), 0.5, 1.5))

    sd = OrderedDict()

    def put_bn(prefix, bn):
        for k in ("weight", "bias", "running_mean", "running_var"):
            sd[prefix + "." + k] = bn[k].astype(np.float32)
        sd[prefix + ".num_batches_tracked"] = np.asarray(1, dtype=np.int64)

    def paired(shp_pairs, scale):
        """[3, 2*npairs, n_out] weights with rows (2i, 2i+1) = (+r_i, -r_i): blind to a uniform offset of the inputs."""
        r = uni(shp_pairs, -1.0, 1.0) * scale
        w = np.zeros((3, 2 * shp_pairs[1], shp_pairs[2]), dtype=np.float64)
        w[:, 0::2, :] = r
        w[:, 1::2, :] = -r
        return w

    def feature_rows(ci, co, signal_in=True):
        """taps[k][ci][co] (float64, post-BN gains) of everything but the signal path's centre tap."""
        w = np.zeros((3, ci, co), dtype=np.float64)
        nf_out, nf_in, np_s, np_f = co - C, ci - C, C // 2, (ci - C) // 2
        if nf_out > 0 and signal_in:
            w[:, :2 * np_s, C:] = paired((3, np_s, nf_out), 0.9 / np.sqrt(3.0 * np_s))
        if nf_out > 0 and np_f > 0:
            w[:, C:C + 2 * np_f, C:] = paired((3, np_f, nf_out), 0.7 / np.sqrt(3.0 * np_f))
        if np_f > 0:
            w[:, C:C + 2 * np_f, :C] = paired((3, np_f, C), coupling / np.sqrt(3.0 * np_f))
        return w

    def solve_conv(prefix, kind, ci, co, bn, in_off, in_gain, out_off, out_gain, first=False, last=False):
        """Weights of one conv (+BN): signal channel j maps  in_off + in_gain*u  ->  out_off + out_gain*u  after BN."""
        w = feature_rows(ci, co, signal_in=not first)      # (the raw pose has no uniform offset to cancel)
        if bn is not None:
            s_bn = bn["weight"] / np.sqrt(bn["running_var"] + BN_EPS)
            beta, mean = bn["bias"], bn["running_mean"]
        else:
            s_bn, beta, mean = np.ones(co), np.zeros(co), np.zeros(co)
        b = np.zeros(co, dtype=np.float64)
        j = np.arange(C)
        if first:          # input is the raw pose x = m + pose_scale*u
            wc = out_gain / (pose_scale * s_bn[:C])
            b[:C] = (out_off - beta[:C]) / s_bn[:C] + mean[:C] - wc * m
        elif last:         # output is the pose itself (no BN, no activation)
            wc = np.full(C, pose_scale / in_gain)
            b[:C] = m - wc * in_off
        else:
            wc = out_gain / (in_gain * s_bn[:C])
            b[:C] = (out_off - beta[:C]) / s_bn[:C] + mean[:C] - wc * in_off
        # every other weight is given as a post-BN gain: divide by the BN scale of its output channel
        w = w / s_bn[None, None, :]
        w[1, j, j] = w[1, j, j] + wc
        b[C:] = (FOFF - beta[C:]) / s_bn[C:] + mean[C:]     # feature outputs: FOFF + zero-mean combination after BN
        if kind == "conv":         # Conv1d weight [co][ci][k], out[t] = sum_k in[t+k-1] w[:,:,k]
            sd[prefix + ".weight"] = np.ascontiguousarray(np.transpose(w, (2, 1, 0))).astype(np.float32)
        else:                      # ConvTranspose1d [ci][co][k] (s=1,p=1): tap k' = 2-k
            sd[prefix + ".weight"] = np.ascontiguousarray(np.transpose(w[::-1], (1, 2, 0))).astype(np.float32)
        sd[prefix + ".bias"] = b.astype(np.float32)

    # ---- encoder
    enc, dec = shape.conv_layers()
    for i, (prefix, kind, ci, co, _) in enumerate(enc):
        bn = bn_params(co)
        solve_conv(prefix + ".0", kind, ci, co, bn, OFF, 1.0, OFF, 1.0, first=(i == 0))
        put_bn(prefix + ".1", bn)
    # ---- Hadamard embedding  Q[d, i] = row_sign[d] * (-1)^popcount(d & col[i]) * col_sign[i] / sqrt(D)
    cols = rng.permutation(D)[:n_sig].astype(np.int64)
    col_sign = (rng.integers(0, 2, size=n_sig) * 2 - 1).astype(np.float64)
    row_sign = (rng.integers(0, 2, size=D) * 2 - 1).astype(np.float64)
    par = np.bitwise_and(np.arange(D, dtype=np.int64)[:, None], cols[None, :])
    for sh in (32, 16, 8, 4, 2, 1):
        par = np.bitwise_xor(par, np.right_shift(par, sh))
    Q = (1.0 - 2.0 * np.bitwise_and(par, 1).astype(np.float64)) * row_sign[:, None] * col_sign[None, :] / np.sqrt(float(D))
    lam = 1.0 / (1.0 + (float(spread) - 1.0) * rng.random(size=n_sig))
    top = shape.hidden[-1]
    # signal index i = t*C + j  <->  flatten index c*T + t with c = j
    t_idx, j_idx = np.divmod(np.arange(n_sig), C)
    flat_idx = j_idx * T + t_idx
    # features -> mu: paired over (channel 2i, 2i+1) at the same frame, so that their offset cancels
    np_f = (top - C) // 2
    w_mu = np.zeros((D, top * T), dtype=np.float64)
    r = uni((D, np_f, T), -1.0, 1.0) * (coupling / (g * np.sqrt(float(2 * np_f * T))))
    w_mu3 = w_mu.reshape(D, top, T)
    w_mu3[:, C:C + 2 * np_f:2, :] = r
    w_mu3[:, C + 1:C + 2 * np_f:2, :] = -r
    w_mu[:, flat_idx] = Q / (g * lam[None, :])
    # mu = Q L^-1 (y - OFF) / g with y = OFF + u on the signal channels: cancel the offset in the bias, accumulated in a
    # fixed order with elementwise adds (a matrix-vector product would be a BLAS reduction)
    b_mu = np.zeros(D, dtype=np.float64)
    for i in range(n_sig):
        b_mu = b_mu - (OFF / (g * lam[i])) * Q[:, i]
    sd["fc_mu.weight"] = w_mu.astype(np.float32)
    sd["fc_mu.bias"] = b_mu.astype(np.float32)
    sd["fc_var.weight"] = (uni((D, top * T), -1.0, 1.0) * (0.05 / np.sqrt(float(top * T)))).astype(np.float32)
    sd["fc_var.bias"] = np.full(D, 2.0 * _log_series(sigma0), dtype=np.float64).astype(np.float32)
    # ---- decoder_input: h[c*T+t] = OFF + (g L Q^T z)[t*C+c] on the signal channels, FOFF + seeded zero-mean rows elsewhere
    w_di = uni((top * T, D), -1.0, 1.0) * (0.7 / np.sqrt(float(D)))
    w_di[flat_idx, :] = (g * lam[:, None]) * Q.T
    b_di = np.full(top * T, FOFF, dtype=np.float64)
    b_di[flat_idx] = OFF
    sd["decoder_input.weight"] = w_di.astype(np.float32)
    sd["decoder_input.bias"] = b_di.astype(np.float32)
    # ---- decoder convs
    for prefix, kind, ci, co, has_bn in dec:
        if prefix == "final_layer.3":
            solve_conv(prefix, kind, ci, co, None, OFF, 1.0, 0.0, 0.0, last=True)
        else:
            bn = bn_params(co)
            solve_conv(prefix + ".0", kind, ci, co, bn, OFF, 1.0, OFF, 1.0)
            put_bn(prefix + ".1", bn)
    return sd


def _log_series(x):
    """ln(x) for 0 < x from +,-,*,/ only (atanh series after scaling by powers of two): bit-identical on every machine."""
    k, y = 0, float(x)
    while y < 0.75:
        y *= 2.0
        k += 1
    while y > 1.5:
        y /= 2.0
        k -= 1
    t = (y - 1.0) / (y + 1.0)
    t2, term, acc = t * t, t, 0.0
    for n in range(40):
        acc += term / (2 * n + 1)
        term *= t2
    LN2 = 0.6931471805599453
    return 2.0 * acc - k * LN2


def state_dict_sha256(state_dict, shape):
    """Exact pin of regenerated weights: SHA-256 over the float32 bytes of every tensor in schema order."""
    import hashlib
    h = hashlib.sha256()
    for name in shape.schema():
        h.update(np.ascontiguousarray(_np(state_dict[name]), dtype=np.float32).tobytes())
    return h.hexdigest()


def state_dict_checksum(state_dict, shape):
    """Order-stable fp64 checksum of a state_dict (used to pin regenerated weights in fixtures)."""
    tot = 0.0
    for k, name in enumerate(shape.schema()):
        a = _np(state_dict[name]).astype(np.float64).ravel()
        tot += float(np.dot(a, np.cos(np.arange(a.size) * 0.37 + k)))
    return tot


def save_checkpoint(path, state_dict):
    """Write `{'state_dict': ...}` like networks/train.py:102-108 (other keys are not read back)."""
    import torch
    sd = OrderedDict((k, torch.from_numpy(np.array(_np(v)))) for k, v in state_dict.items())
    torch.save({"epoch": 19, "state_dict": sd}, path)


def load_checkpoint_file(path, trust=None):
    """The whole dict of a checkpoint file (networks/train.py:102-108: epoch, args, state_dict, eval_result, optimizer) through
    torch's restricted unpickler: tensors, containers, primitives and numpy scalars / arrays only -- a downloaded file cannot run
    code.  A file that needs more (the reference never writes one) is refused unless `trust=True` or GEM_TRUST_CHECKPOINTS=1."""
    import os
    import pickle
    import torch
    if trust is None:
        trust = os.environ.get("GEM_TRUST_CHECKPOINTS") == "1"
    allow = [np.dtype, np.ndarray]
    for name in (("_core",) if hasattr(np, "_core") else ("core",)):          # numpy 2.x / 1.x module layout
        mod = getattr(np, name, None)
        ma = getattr(mod, "multiarray", None)
        for fn in ("scalar", "_reconstruct"):
            if ma is not None and hasattr(ma, fn):
                allow.append(getattr(ma, fn))
    allow += [type(np.dtype(t)) for t in ("float32", "float64", "int64", "int32", "bool")]
    # the same two functions under the module path of the OTHER numpy generation: the reference's train.py stores
    # eval_result = np.mean(...), a numpy scalar, so a checkpoint written under numpy 1.x names numpy.core.multiarray.scalar and
    # one written under 2.x numpy._core.multiarray.scalar -- whichever numpy reads it (torch matches the full name)
    for legacy in ("numpy.core.multiarray", "numpy._core.multiarray"):
        for fn in ("scalar", "_reconstruct"):
            target = next((a for a in allow if getattr(a, "__name__", None) == fn), None)
            if target is not None:
                allow.append((target, "%s.%s" % (legacy, fn)))
    try:
        with torch.serialization.safe_globals(allow):
            return torch.load(path, map_location="cpu", weights_only=True)
    except pickle.UnpicklingError as e:
        if not trust:
            raise pickle.UnpicklingError("%s holds objects the restricted unpickler refuses (%s); pass trust=True / set "
                                         "GEM_TRUST_CHECKPOINTS=1 only for files whose origin you trust" % (path, str(e).splitlines()[0])) from e
    return torch.load(path, map_location="cpu", weights_only=False)


def load_checkpoint(path, trust=None):
    """`torch.load(path)['state_dict']` (optimizer.py:59), through the restricted unpickler (load_checkpoint_file)."""
    return load_checkpoint_file(path, trust)["state_dict"]
