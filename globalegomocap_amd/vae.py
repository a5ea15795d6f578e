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


def load_checkpoint(path):
    """`torch.load(path)['state_dict']` (optimizer.py:59)."""
    import torch
    return torch.load(path, map_location="cpu", weights_only=False)["state_dict"]
