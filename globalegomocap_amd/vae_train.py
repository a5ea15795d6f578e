"""Training / fine-tuning of the motion VAE on the device.

Host-side mirror of the reference's `networks/train.py:36-127` (class `Train`) over `gem_trainer_*` (include/gem_hip.h,
csrc/train.hip): one `VAETrainer.step` is the loop body `zero_grad -> forward (train mode) -> loss_function -> backward ->
Adam.step` (train.py:77-83), `fit` the epoch loop with the same shuffling DataLoader semantics (shuffle, drop_last), the same
M_N = kl_weight * batch_size / len(dataset), the running-loss log every `log_step` steps, the evaluation pass
(train.py:110-123: eval-mode reconstruction MPJPE) and the per-epoch checkpoint `{'epoch', 'state_dict', 'eval_result', ...}`
(train.py:102-108) in the reference's own schema, which `optimizer.py:59-60` and `WindowEngine.load_vae` read back.

The arena layout functions at the top are numpy-only (tested on the CPU); everything that computes goes through the HIP library
and fails loudly without it.
"""
import ctypes as C
import os
from collections import OrderedDict

import numpy as np

from .vae import VAEShape, check_state_dict, _np

PAD = 64


def _pad(x):
    return (x + PAD - 1) // PAD * PAD


def arena_layout(shape):
    """The packed arena of gem_trainer (include/gem_hip.h): ([(key, kind, offset, (dims...))...], n_params, [(key, offset, N, C)...],
    n_stats).  kinds: conv / convT (weight [3][N][K]), vec (length N, real C), fc (stacked [2Dp][T*topp]), fc_b, din, din_b."""
    T, D, Dp = shape.seq_len, shape.latent_dim, _pad(shape.latent_dim)
    top, topp = shape.hidden[-1], _pad(shape.hidden[-1])
    enc, dec = shape.conv_layers()
    items, stats, off, soff = [], [], 0, 0

    def conv(prefix, kind, ci, co, bn):
        nonlocal off, soff
        K, N = _pad(ci), _pad(co)
        wkey = prefix + (".0.weight" if bn else ".weight")
        items.append((wkey, kind, off, (N, K, co, ci)))
        off += 3 * N * K
        items.append((prefix + (".0.bias" if bn else ".bias"), "vec", off, (N, co)))
        off += N
        if bn:
            items.append((prefix + ".1.weight", "vec", off, (N, co)))
            off += N
            items.append((prefix + ".1.bias", "vec", off, (N, co)))
            off += N
            stats.append((prefix + ".1.running_mean", soff, N, co))
            stats.append((prefix + ".1.running_var", soff + N, N, co))
            soff += 2 * N

    for prefix, kind, ci, co, bn in enc:
        conv(prefix, kind, ci, co, bn)
    items.append(("fc", "fc", off, (Dp, T, topp, D, top)))
    off += 2 * Dp * T * topp
    items.append(("fc.bias", "fc_b", off, (Dp, D)))
    off += 2 * Dp
    items.append(("decoder_input.weight", "din", off, (T, topp, Dp, top, D)))
    off += T * topp * Dp
    items.append(("decoder_input.bias", "din_b", off, (T, topp, top)))
    off += T * topp
    for prefix, kind, ci, co, bn in dec:
        conv(prefix, kind, ci, co, bn)
    return items, off, stats, soff


def pack_arena(state, shape):
    """Reference-schema state_dict -> (params arena, statistics arena), fp32."""
    check_state_dict(state, shape)
    items, n, stats, ns = arena_layout(shape)
    P, S = np.zeros(n, np.float32), np.zeros(ns, np.float32)
    T = shape.seq_len
    for key, kind, off, dims in items:
        if kind in ("conv", "convT"):
            N, K, co, ci = dims
            w = _np(state[key]).astype(np.float32)
            taps = np.transpose(w, (2, 0, 1)) if kind == "conv" else np.transpose(w[:, :, ::-1], (2, 1, 0))    # [tap][n][k]
            blk = np.zeros((3, N, K), np.float32)
            blk[:, :co, :ci] = taps
            P[off:off + blk.size] = blk.ravel()
        elif kind == "vec":
            N, co = dims
            P[off:off + co] = _np(state[key]).astype(np.float32)
        elif kind == "fc":
            Dp, _, topp, D, top = dims
            blk = np.zeros((2, Dp, T, topp), np.float32)
            for i, name in enumerate(("fc_mu", "fc_var")):
                w = _np(state[name + ".weight"]).astype(np.float32).reshape(D, top, T)
                blk[i, :D, :, :top] = np.transpose(w, (0, 2, 1))
            P[off:off + blk.size] = blk.ravel()
        elif kind == "fc_b":
            Dp, D = dims
            P[off:off + D] = _np(state["fc_mu.bias"])
            P[off + Dp:off + Dp + D] = _np(state["fc_var.bias"])
        elif kind == "din":
            _, topp, Dp, top, D = dims
            w = _np(state[key]).astype(np.float32).reshape(top, T, D)
            blk = np.zeros((T, topp, Dp), np.float32)
            blk[:, :top, :D] = np.transpose(w, (1, 0, 2))
            P[off:off + blk.size] = blk.ravel()
        elif kind == "din_b":
            _, topp, top = dims
            blk = np.zeros((T, topp), np.float32)
            blk[:, :top] = _np(state[key]).astype(np.float32).reshape(top, T).T
            P[off:off + blk.size] = blk.ravel()
    for key, off, N, co in stats:
        S[off:off + N] = 1.0 if key.endswith("running_var") else 0.0
        S[off:off + co] = _np(state[key]).astype(np.float32)
    return P, S


def unpack_arena(P, shape, S=None):
    """Inverse of pack_arena for a parameter-shaped arena (parameters, gradients, Adam moments): OrderedDict in schema order
    (statistics keys only when `S` is given)."""
    items, n, stats, ns = arena_layout(shape)
    P = np.asarray(P, np.float32)
    if P.size != n:
        raise ValueError("arena has %d floats, the layout %d" % (P.size, n))
    T = shape.seq_len
    out = {}
    for key, kind, off, dims in items:
        if kind in ("conv", "convT"):
            N, K, co, ci = dims
            taps = P[off:off + 3 * N * K].reshape(3, N, K)[:, :co, :ci]
            out[key] = np.ascontiguousarray(np.transpose(taps, (1, 2, 0)) if kind == "conv" else np.transpose(taps, (2, 1, 0))[:, :, ::-1])
        elif kind == "vec":
            out[key] = P[off:off + dims[1]].copy()
        elif kind == "fc":
            Dp, _, topp, D, top = dims
            blk = P[off:off + 2 * Dp * T * topp].reshape(2, Dp, T, topp)
            for i, name in enumerate(("fc_mu", "fc_var")):
                out[name + ".weight"] = np.ascontiguousarray(np.transpose(blk[i, :D, :, :top], (0, 2, 1))).reshape(D, top * T)
        elif kind == "fc_b":
            Dp, D = dims
            out["fc_mu.bias"], out["fc_var.bias"] = P[off:off + D].copy(), P[off + Dp:off + Dp + D].copy()
        elif kind == "din":
            _, topp, Dp, top, D = dims
            blk = P[off:off + T * topp * Dp].reshape(T, topp, Dp)[:, :top, :D]
            out[key] = np.ascontiguousarray(np.transpose(blk, (1, 0, 2))).reshape(top * T, D)
        elif kind == "din_b":
            _, topp, top = dims
            out[key] = np.ascontiguousarray(P[off:off + T * topp].reshape(T, topp)[:, :top].T).reshape(top * T)
    if S is not None:
        S = np.asarray(S, np.float32)
        if S.size != ns:
            raise ValueError("statistics arena has %d floats, the layout %d" % (S.size, ns))
        for key, off, N, co in stats:
            out[key] = S[off:off + co].copy()
    return OrderedDict((k, out[k]) for k in shape.schema() if k in out)


def initial_state_dict(shape, seed=0):
    """ConvVAE.__init__ (SeqConvVAE.py:11-92) parameter initialisation -- torch's defaults for Conv1d / ConvTranspose1d / Linear
    (kaiming_uniform(a=sqrt(5)) = U(+-1/sqrt(fan_in)) for weights and biases), BatchNorm weight 1 / bias 0 / mean 0 / var 1 --
    drawn from a numpy generator (the values of torch's own RNG stream are not part of the reference's behaviour)."""
    rng = np.random.default_rng(seed)
    sd = OrderedDict()
    for name, shp in shape.schema().items():
        # BatchNorm tensors are "<block>.1.{weight,bias,running_mean,running_var}" (a conv is "<block>.0.*", the last conv "final_layer.3.*")
        if name.endswith("running_mean") or name.endswith(".1.bias"):
            sd[name] = np.zeros(shp, np.float32)
        elif name.endswith("running_var") or name.endswith(".1.weight"):
            sd[name] = np.ones(shp, np.float32)
        elif name.endswith(".weight"):
            # torch's fan_in = size(1) * receptive field, also for ConvTranspose1d's [C_in, C_out, k] (i.e. C_out * k there)
            fan_in = shp[1] * (shp[2] if len(shp) == 3 else 1)
            b = 1.0 / np.sqrt(fan_in)
            sd[name] = rng.uniform(-b, b, shp).astype(np.float32)
            sd["__fan_in__" + name] = fan_in
        else:
            w = name[:-len("bias")] + "weight"
            b = 1.0 / np.sqrt(sd["__fan_in__" + w])
            sd[name] = rng.uniform(-b, b, shp).astype(np.float32)
    return OrderedDict((k, v) for k, v in sd.items() if not k.startswith("__fan_in__"))


class VAETrainer:
    """`Train` of networks/train.py over the HIP library: Adam(lr, weight_decay) on ConvVAE.loss_function, BatchNorm in train mode."""

    def __init__(self, shape=None, batch_size=64, lr=1e-4, weight_decay=0.0, betas=(0.9, 0.999), eps=1e-8, bn_momentum=0.1,
                 recon_reduction="mean", device=None, state_dict=None, seed=0):
        import torch
        from . import _capi
        from .engine import N_JOINTS
        self.lib = _capi.load_library()
        if not torch.cuda.is_available():
            raise _capi.GemError("no HIP device visible: the VAE trainer has no CPU path")
        self.shape = shape or VAEShape()
        if self.shape.channels != 3 * N_JOINTS:
            raise ValueError("the trainer is built for %d-joint poses" % N_JOINTS)
        if device is not None and not isinstance(device, int):
            device = torch.device(device).index
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.batch_size = int(batch_size)
        cfg = _capi.GemConfig()
        cfg.seq_len, cfg.n_joints, cfg.latent_dim = self.shape.seq_len, N_JOINTS, self.shape.latent_dim
        cfg.n_hidden = len(self.shape.hidden)
        for i, v in enumerate(self.shape.hidden):
            cfg.hidden[i] = v
        cfg.heat_h = cfg.heat_w = 64
        cfg.n_poly, cfg.poly[0] = 1, 1.0
        cfg.max_windows, cfg.device = self.batch_size, self.device.index
        self._t = C.c_void_p()
        _capi.check(self.lib.gem_trainer_create(C.byref(cfg), C.byref(self._t)), self.lib)
        import weakref
        self._finalizer = weakref.finalize(self, VAETrainer._destroy, self.lib, self._t)
        n, ns = C.c_int64(), C.c_int64()
        _capi.check(self.lib.gem_trainer_sizes(self._t, C.byref(n), C.byref(ns)), self.lib)
        self.n_params, self.n_stats = n.value, ns.value
        lay = arena_layout(self.shape)
        if (lay[1], lay[3]) != (self.n_params, self.n_stats):
            raise _capi.GemError("arena layout of the library (%d, %d) and of vae_train.py (%d, %d) differ"
                                 % (self.n_params, self.n_stats, lay[1], lay[3]))
        self.opts = _capi.GemTrainOpts(lr=float(lr), beta1=float(betas[0]), beta2=float(betas[1]), eps=float(eps),
                                       weight_decay=float(weight_decay), kld_weight=0.0, bn_momentum=float(bn_momentum),
                                       recon_sum=1 if recon_reduction == "sum" else 0, reserved=0)
        self.steps = 0
        self.forwards = 0
        self._fused_last = False         # the last step ran in the training-loop mode (gem_trainer_step update = 2)
        self._grad = None
        self._losses = torch.zeros(3, dtype=torch.float64, device=self.device)
        self._gen = torch.Generator(device=self.device)
        self._gen.manual_seed(seed)
        self.load_state_dict(state_dict if state_dict is not None else initial_state_dict(self.shape, seed))

    def close(self):
        """Free the device memory.  Also runs (through weakref.finalize) when the trainer is collected or -- before the HIP runtime
        is torn down -- at interpreter exit: destroying a handle from __del__ during shutdown can hang under a profiler."""
        self._finalizer()

    @staticmethod
    def _destroy(lib, handle):
        if handle.value:
            lib.gem_trainer_destroy(handle)
            handle.value = None

    # ---- parameters
    def _up(self, what, arr):
        from . import _capi
        a = np.ascontiguousarray(arr, np.float32)
        _capi.check(self.lib.gem_trainer_upload(self._t, what, a.ctypes.data_as(C.c_void_p), a.size), self.lib)

    def _down(self, what):
        from . import _capi
        a = np.empty(self.n_stats if what == 2 else self.n_params, np.float32)
        _capi.check(self.lib.gem_trainer_download(self._t, what, a.ctypes.data_as(C.c_void_p), a.size), self.lib)
        return a

    def load_state_dict(self, state):
        """network.load_state_dict: parameters, running statistics and BatchNorm's num_batches_tracked (when the dict carries it);
        Adam's moments and step count start over (load_optimizer_state restores them)."""
        from . import _capi
        P, S = pack_arena(state, self.shape)
        self._up(0, P)
        self._up(2, S)
        self._up(3, np.zeros_like(P))
        self._up(4, np.zeros_like(P))
        self.steps = 0
        _capi.check(self.lib.gem_trainer_set_step(self._t, 0), self.lib)
        nbt = [int(np.asarray(v)) for k, v in state.items() if k.endswith("num_batches_tracked")]
        self.forwards = nbt[0] if nbt else 0

    def _param_keys(self):
        """Keys of network.parameters() in registration order (the reference's ConvVAE: SeqConvVAE.py:11-92) = the state_dict order
        without the BatchNorm buffers."""
        return [k for k in self.shape.schema() if not (k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked"))]

    def torch_optimizer_state_dict(self):
        """`torch.optim.Adam.state_dict()` of the reference's optimizer (networks/train.py:102-108 saves exactly that): 'state'
        indexed by parameter position with 'step' / 'exp_avg' / 'exp_avg_sq', one entry in 'param_groups'.
        `torch.optim.Adam(network.parameters()).load_state_dict(...)` accepts it."""
        import torch
        opt = self.optimizer_state()
        keys = self._param_keys()
        state = {i: {"step": torch.tensor(float(opt["step"])), "exp_avg": torch.from_numpy(np.array(opt["exp_avg"][k])),
                     "exp_avg_sq": torch.from_numpy(np.array(opt["exp_avg_sq"][k]))} for i, k in enumerate(keys)} if opt["step"] > 0 else {}
        group = {"lr": self.opts.lr, "betas": (self.opts.beta1, self.opts.beta2), "eps": self.opts.eps, "weight_decay": self.opts.weight_decay,
                 "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "decoupled_weight_decay": False, "params": list(range(len(keys)))}
        return {"state": state, "param_groups": [group]}

    def load_optimizer_state(self, opt_sd):
        """Resume: Adam's moments and step count from `torch.optim.Adam.state_dict()` (the reference's checkpoints, and this
        class's own), or from this class's `optimizer_state()` dict."""
        from . import _capi
        keys = self._param_keys()
        if "state" in opt_sd:
            st = opt_sd["state"]
            if not st:
                step, m, v = 0, {}, {}
            else:
                step = int(round(float(np.asarray(st[0]["step"]))))
                m = {k: np.asarray(st[i]["exp_avg"], np.float32) for i, k in enumerate(keys)}
                v = {k: np.asarray(st[i]["exp_avg_sq"], np.float32) for i, k in enumerate(keys)}
            g = opt_sd["param_groups"][0]
            self.opts.lr, self.opts.eps, self.opts.weight_decay = float(g["lr"]), float(g["eps"]), float(g["weight_decay"])
            self.opts.beta1, self.opts.beta2 = float(g["betas"][0]), float(g["betas"][1])
        else:
            step = int(opt_sd["step"])
            m = {k: np.asarray(x, np.float32) for k, x in opt_sd["exp_avg"].items()}
            v = {k: np.asarray(x, np.float32) for k, x in opt_sd["exp_avg_sq"].items()}
        zeros = {k: np.zeros(shp, np.float32) for k, shp in self.shape.schema().items()}
        for what, d in ((3, m), (4, v)):
            full = dict(zeros)
            full.update(d)
            self._up(what, pack_arena(full, self.shape)[0])
        self.steps = step
        _capi.check(self.lib.gem_trainer_set_step(self._t, step), self.lib)

    def load_checkpoint(self, path, trust=None):
        """A checkpoint of `fit` (or of the reference's networks/train.py:102-108): weights, statistics, Adam state.  Returns the
        epoch it was written after.  Read through torch's restricted unpickler (vae.load_checkpoint_file)."""
        from .vae import load_checkpoint_file
        ck = load_checkpoint_file(path, trust)
        self.load_state_dict({k: np.asarray(v) for k, v in ck["state_dict"].items()})
        if ck.get("optimizer"):
            self.load_optimizer_state(ck["optimizer"])
        return int(ck.get("epoch", 0))

    def state_dict(self):
        """network.state_dict() in the reference's schema (numpy arrays; incl. num_batches_tracked like torch's BatchNorm)."""
        sd = unpack_arena(self._down(0), self.shape, self._down(2))
        out = OrderedDict()
        for k, v in sd.items():
            out[k] = v
            if k.endswith("running_var"):
                # (torch counts FORWARD passes in train mode, not optimiser steps: gradient-only passes count too)
                out[k[:-len("running_var")] + "num_batches_tracked"] = np.array(self.forwards, np.int64)
        return out

    def gradients(self):
        """The gradients of the last step, keyed like the parameters (p.grad after loss.backward()).  After a training-loop step
        (keep_gradients=False) the two linear layers' weight gradients were never written out: those keys are absent (the library
        hands their arena ranges back as NaN), never the stale values of an earlier step."""
        g = unpack_arena(self._down(1), self.shape)
        return OrderedDict((k, v) for k, v in g.items() if not np.isnan(np.asarray(v)).all()) if self._fused_last else g

    def optimizer_state(self):
        """Adam's exp_avg / exp_avg_sq per parameter key and the step count (optimizer.state_dict())."""
        return {"step": self.steps, "exp_avg": unpack_arena(self._down(3), self.shape), "exp_avg_sq": unpack_arena(self._down(4), self.shape)}

    # ---- one step (train.py:77-83)
    def step(self, poses, kld_weight, eps=None, update=True, sync=True, keep_gradients=True):
        """poses [B,T,45] (device or host); eps [B,D] or None (drawn on the device like torch.randn_like).  Returns
        (loss, recon_loss, kld_loss) as floats (sync=True) or the device tensor that will hold them.

        keep_gradients=False (with update): the training loop's mode -- the two linear layers (97 % of the parameters) form their
        weight gradient inside their Adam step (gem_trainer_step update = 2) and do not leave it in the gradient arena, so
        `gradients()` is not valid for them afterwards; the parameters, moments and losses are those of the default mode."""
        import torch
        from . import _capi
        x = torch.as_tensor(poses, dtype=torch.float32, device=self.device).contiguous()
        B = int(x.shape[0])
        if tuple(x.shape[1:]) != (self.shape.seq_len, self.shape.channels):
            raise ValueError("poses must be [B,%d,%d]" % (self.shape.seq_len, self.shape.channels))
        if eps is None:
            e = torch.randn(B, self.shape.latent_dim, device=self.device, dtype=torch.float32, generator=self._gen)
        else:
            e = torch.as_tensor(eps, dtype=torch.float32, device=self.device).contiguous()
            if tuple(e.shape) != (B, self.shape.latent_dim):
                raise ValueError("eps must be [B,%d]" % self.shape.latent_dim)
        self.opts.kld_weight = float(kld_weight)
        s = torch.cuda.current_stream(self.device).cuda_stream
        _capi.check(self.lib.gem_trainer_step(self._t, B, x.data_ptr(), e.data_ptr(), C.byref(self.opts),
                                              (1 if keep_gradients else 2) if update else 0, self._losses.data_ptr(), C.c_void_p(s)), self.lib)
        self.forwards += 1               # every train-mode forward updates the running statistics (like torch's BatchNorm)
        self._fused_last = bool(update) and not keep_gradients
        if update:
            self.steps += 1
        if not sync:
            return self._losses
        return tuple(float(v) for v in self._losses.cpu())

    # ---- data parallel (one process per GPU; the reference itself trains on one device)
    def arena_tensor(self, what=1):
        """Arena `what` (0 parameters, 1 gradients, 2 statistics, 3 / 4 Adam moments) as a torch tensor over the library's memory."""
        import torch
        from . import _capi
        ptr, n = C.c_void_p(), C.c_int64()
        _capi.check(self.lib.gem_trainer_arena(self._t, what, C.byref(ptr), C.byref(n)), self.lib)

        class _View:
            __cuda_array_interface__ = {"shape": (n.value,), "typestr": "<f4", "data": (ptr.value, False), "version": 2}
        return torch.as_tensor(_View(), device=self.device)

    def step_data_parallel(self, poses, kld_weight, eps=None, group=None, sync=True):
        """One step of DistributedDataParallel-style training: this rank's batch -> gradients (gem_trainer_step, update = 0),
        ONE all-reduce of the flat gradient arena (RCCL through torch.distributed; `group` = None: the default group), Adam on
        the mean gradient (gem_trainer_apply with grad_scale = 1 / world size).  BatchNorm statistics stay per rank (DDP without
        SyncBatchNorm).  Returns the losses averaged over the ranks."""
        import torch
        import torch.distributed as dist
        from . import _capi
        world = dist.get_world_size(group)
        losses = self.step(poses, kld_weight, eps=eps, update=False, sync=False)
        if getattr(self, "_grad", None) is None:
            self._grad = self.arena_tensor(1)
        dist.all_reduce(self._grad, group=group)
        s = torch.cuda.current_stream(self.device).cuda_stream
        _capi.check(self.lib.gem_trainer_apply(self._t, C.byref(self.opts), 1.0 / world, C.c_void_p(s)), self.lib)
        self.steps += 1
        mean = losses.clone()
        dist.all_reduce(mean, group=group)
        mean /= world
        return tuple(float(v) for v in mean.cpu()) if sync else mean

    def broadcast_parameters(self, src=0, group=None):
        """Every rank starts from rank `src`'s parameters, statistics and Adam state (DDP's initial broadcast)."""
        import torch.distributed as dist
        for what in (0, 2, 3, 4):
            dist.broadcast(self.arena_tensor(what), src=src, group=group)

    # ---- the epoch loop (train.py:65-108) and the evaluation pass (train.py:110-127)
    def evaluate(self, windows, batch_size=None):
        """Eval-mode reconstruction MPJPE (train.py:110-127): a WindowEngine with the current weights encodes with the random
        reparameterisation of ConvVAE.forward and decodes; the mean over batches of the per-batch mean joint distance."""
        import torch
        from .engine import WindowEngine
        bs = int(batch_size or self.batch_size)
        eng = WindowEngine(self.shape, max_windows=bs, device=self.device.index)
        try:
            eng.load_vae(0, self.state_dict())
            data = torch.as_tensor(np.asarray(windows), dtype=torch.float32)
            errs = []
            for i in range(0, data.shape[0], bs):
                x = data[i:i + bs].to(self.device)
                e = torch.randn(x.shape[0], self.shape.latent_dim, device=self.device, generator=self._gen)
                z = eng.encode(0, x, eps=e)[2]
                rec = eng.decode(0, z)
                d = (rec.reshape(x.shape[0], self.shape.seq_len, -1, 3) - x.reshape(x.shape[0], self.shape.seq_len, -1, 3)).norm(dim=-1)
                errs.append(float(d.mean()))
            return float(np.mean(errs))
        finally:
            eng.close()

    def fit(self, train_windows, epochs=20, kl_weight=0.25, test_windows=None, log_step=100, checkpoint_dir=None, seed=0, log=print,
            args=None, group=None, data_parallel=False):
        """Train.train(): `epochs` passes over a shuffled, drop_last DataLoader of `train_windows` [n,T,45].

        data_parallel=True (torch.distributed initialised, one process per GPU): every rank walks the same permutation (same
        seed) and takes the batches rank, rank + world, ... (a DistributedSampler's split); gradients are averaged over the
        ranks every step; rank 0 logs, evaluates and writes the checkpoints."""
        import torch
        from .vae import save_checkpoint  # noqa: F401  (same schema; the file below carries train.py's extra keys)
        data = torch.as_tensor(np.asarray(train_windows), dtype=torch.float32, device=self.device)
        n, bs = int(data.shape[0]), self.batch_size
        if n < bs:
            raise ValueError("the dataset (%d windows) is smaller than one batch (%d): drop_last leaves nothing" % (n, bs))
        m_n = float(kl_weight) * bs / n
        g = torch.Generator(device="cpu").manual_seed(seed)
        running = torch.zeros(3, dtype=torch.float64, device=self.device)
        count, history = 0, []
        rank, world = 0, 1
        if data_parallel:
            import torch.distributed as dist
            rank, world = dist.get_rank(group), dist.get_world_size(group)
            self.broadcast_parameters(0, group)
            if rank != 0:
                log = lambda *a, **k: None          # noqa: E731  (rank 0 reports, like a DDP training script)
        for e in range(int(epochs)):
            perm = torch.randperm(n, generator=g).to(self.device)
            for i in range(rank, (n // bs) // world * world, world):
                if data_parallel:
                    running += self.step_data_parallel(data[perm[i * bs:(i + 1) * bs]], m_n, group=group, sync=False)
                else:
                    running += self.step(data[perm[i * bs:(i + 1) * bs]], m_n, sync=False, keep_gradients=False)
                if count % log_step == 0 and count != 0:
                    r = running.cpu()
                    log("running loss is: {}".format(float(r[0])))
                    log("running recon loss is: {}".format(float(r[1])))
                    history.append((count, float(r[0]), float(r[1])))
                    running.zero_()
                count += 1
            if rank != 0:
                continue
            eval_loss = self.evaluate(test_windows if test_windows is not None else train_windows) if test_windows is not False else None
            if eval_loss is not None:
                log("eval loss is: {}".format(eval_loss))
            if checkpoint_dir is not None:
                os.makedirs(checkpoint_dir, exist_ok=True)
                sd = OrderedDict((k, torch.from_numpy(np.array(v))) for k, v in self.state_dict().items())
                # (train.py:102-108: 'optimizer' is torch.optim.Adam.state_dict(); same schema here, so either side resumes the other's file)
                torch.save({"epoch": e + 1, "args": dict(args or {}), "state_dict": sd, "eval_result": eval_loss,
                            "optimizer": self.torch_optimizer_state_dict()},
                           os.path.join(checkpoint_dir, str(e) + ".pth.tar"))
        return history


def fit_vae_device(shape, windows, steps=2000, batch=128, lr=2e-3, kl_weight=0.01, seed=0, device=None, latent_gain=1.0):
    """`vae_torch.fit_vae` on the device trainer: what SURVEY 8 f.4 is for -- producing well-conditioned weights in the reference's
    checkpoint schema for accuracy experiments.  Same recipe (Adam on summed squared error + kl_weight * KLD over random batches of
    synthetic windows [n,T,45], one-cycle learning rate, output bias started at the mean pose), returns (state_dict, reconstruction
    error in m); the post-hoc latent gauge `latent_gain` is applied like there."""
    import math
    import torch
    data = torch.as_tensor(np.asarray(windows), dtype=torch.float32)
    init = initial_state_dict(shape, seed)
    init["final_layer.3.bias"] = data.mean(dim=(0, 1)).numpy().astype(np.float32)
    init["final_layer.3.weight"] = init["final_layer.3.weight"] * np.float32(0.1)
    tr = VAETrainer(shape, batch_size=batch, lr=lr, recon_reduction="sum", device=device, state_dict=init, seed=seed)
    try:
        data = data.to(tr.device)
        g = torch.Generator(device="cpu").manual_seed(seed)
        # torch.optim.lr_scheduler.OneCycleLR defaults: 30 % cosine warm-up from max_lr / 25, cosine decay to max_lr / 25e4
        up = float(int(0.3 * steps) - 1)
        # (every step's batch indices in ONE upload: a per-step host-to-device copy is a synchronisation point that keeps the host from
        # enqueueing ahead of the 0.8 ms steps)
        idx_all = torch.randint(0, data.shape[0], (steps, batch), generator=g).to(tr.device)
        for it in range(steps):
            if it <= up:
                tr.opts.lr = lr / 25 + (lr - lr / 25) * 0.5 * (1 - math.cos(math.pi * it / max(up, 1.0)))
            else:
                tr.opts.lr = lr / 25e4 + (lr - lr / 25e4) * 0.5 * (1 + math.cos(math.pi * (it - up) / max(steps - 1 - up, 1.0)))
            tr.step(data[idx_all[it]], kl_weight, sync=False, keep_gradients=False)
        sd = tr.state_dict()
        from .engine import WindowEngine
        eng = WindowEngine(shape, max_windows=256, device=tr.device.index)
        try:
            eng.load_vae(0, sd)
            x = data[:256]
            rec = eng.decode(0, eng.encode(0, x)[0])
            err = float((rec.reshape(-1, 15, 3) - x.reshape(-1, 15, 3)).norm(dim=-1).mean())
        finally:
            eng.close()
    finally:
        tr.close()
    sd = OrderedDict((k, torch.from_numpy(np.array(v))) for k, v in sd.items() if not k.endswith("num_batches_tracked"))
    if latent_gain != 1.0:
        gn = float(latent_gain)
        sd["decoder_input.weight"] = sd["decoder_input.weight"] * gn
        sd["fc_mu.weight"], sd["fc_mu.bias"] = sd["fc_mu.weight"] / gn, sd["fc_mu.bias"] / gn
        sd["fc_var.bias"] = sd["fc_var.bias"] - 2.0 * float(np.log(gn))
    return sd, err


def _cli():
    """`python -m globalegomocap_amd.vae_train`: the flags of networks/config.py:5-49 that `Train` reads, over a window file.

    --train_data_path takes a .npy / .npz ('windows') of [n, seq_length, 45] relative-global pose windows (what AMASSDataset's
    __getitem__ yields, networks/dataset/global_dataset.py:34-38; building them from AMASS pickles is data preparation and stays
    with the reference) or `synthetic:<n>` for synthetic motion.  Checkpoints go to logs/<log_dir>/checkpoints/<e>.pth.tar
    (train.py:19-33,102-108)."""
    import argparse
    import datetime
    from . import synth
    p = argparse.ArgumentParser(description="Train the motion VAE on MI355X (mirror of networks/train.py)")
    p.add_argument("--train_data_path", required=True)
    p.add_argument("--test_data_path", default=None)
    p.add_argument("--latent_dim", type=int, required=True)
    p.add_argument("--seq_length", type=int, required=True)
    p.add_argument("--kl_weight", type=float, required=True)
    p.add_argument("--epoch", type=int, default=20)
    p.add_argument("--batch_size", type=int, default=64)
    p.add_argument("--learning_rate", type=float, default=1e-4)
    p.add_argument("--weight_decay", type=float, default=0.0)
    p.add_argument("--log_dir", default=None)
    p.add_argument("--log_step", type=int, default=100)
    p.add_argument("--seed", type=int, default=0)
    a = p.parse_args()

    def load(path, seed):
        if path.startswith("synthetic:"):
            return synth.make_training_windows(int(path.split(":", 1)[1]), a.seq_length, seed)
        d = np.load(path)
        return np.asarray(d["windows"] if hasattr(d, "files") else d, np.float32).reshape(-1, a.seq_length, 45)
    train = load(a.train_data_path, a.seed)
    test = load(a.test_data_path, a.seed + 1) if a.test_data_path else train[-min(len(train), 10 * a.batch_size):]
    log_dir = os.path.join("logs", a.log_dir or datetime.datetime.now().strftime("%m.%d-%H:%M:%S"))
    print("making save dir at: {}".format(log_dir))
    shape = VAEShape(latent_dim=a.latent_dim, seq_len=a.seq_length)
    tr = VAETrainer(shape, batch_size=a.batch_size, lr=a.learning_rate, weight_decay=a.weight_decay, seed=a.seed)
    try:
        print("---------------------Start Training-----------------------")
        tr.fit(train, epochs=a.epoch, kl_weight=a.kl_weight, test_windows=test, log_step=a.log_step,
               checkpoint_dir=os.path.join(log_dir, "checkpoints"), seed=a.seed, args=vars(a))
    finally:
        tr.close()


if __name__ == "__main__":
    _cli()
