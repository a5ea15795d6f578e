"""torch.nn twin of the motion VAE in the reference's checkpoint schema -- for PRODUCING weights only.

The window optimiser never touches this module (the decoder/encoder it runs are the HIP kernels).
It exists because the trained checkpoints are an external download (`README.md:25`): tests and
`bench.py` fit a VAE briefly on synthetic motion so that decode(encode(x)) ~ x and the L-BFGS
problem is as well conditioned as with real weights, and save it with the
key names of `ConvVAE` (`networks/models/SeqConvVAE.py:11-92`) so that either implementation can
load it.  Training objective = the reference's `loss_function` (SeqConvVAE.py:213-219).
"""
import numpy as np
import torch
from torch import nn
import torch.nn.functional as F


def _block(conv, c_out):
    return nn.Sequential(conv, nn.BatchNorm1d(c_out), nn.LeakyReLU())


class MotionVAE(nn.Module):
    """Same layer graph and state_dict keys as the reference ConvVAE (with_bone_length=False)."""

    def __init__(self, latent_dim=2048, seq_len=10, hidden=(64, 64, 128, 256, 512), channels=45):
        super().__init__()
        self.seq_len, self.top, self.latent_dim = seq_len, hidden[-1], latent_dim
        dims = [channels] + list(hidden)
        self.encoder = nn.Sequential(*[_block(nn.Conv1d(a, b, 3, padding=1), b) for a, b in zip(dims, dims[1:])])
        self.fc_mu = nn.Linear(self.top * seq_len, latent_dim)
        self.fc_var = nn.Linear(self.top * seq_len, latent_dim)
        self.decoder_input = nn.Linear(latent_dim, self.top * seq_len)
        rev = list(reversed(hidden))
        self.decoder = nn.Sequential(*[_block(nn.ConvTranspose1d(a, b, 3, padding=1), b) for a, b in zip(rev, rev[1:])])
        last = rev[-1]
        self.final_layer = nn.Sequential(nn.ConvTranspose1d(last, last, 3, padding=1), nn.BatchNorm1d(last),
                                         nn.LeakyReLU(), nn.Conv1d(last, channels, 3, padding=1))

    def moments(self, pose):                       # pose [B,T,45]
        h = self.encoder(pose.permute(0, 2, 1).contiguous()).flatten(1)
        return self.fc_mu(h), self.fc_var(h)

    def latent(self, pose, eps):
        mu, logvar = self.moments(pose)
        return eps * torch.exp(0.5 * logvar) + mu

    def decode_raw(self, z):                       # -> [B,45,T]
        h = self.decoder_input(z).view(-1, self.top, self.seq_len)
        return self.final_layer(self.decoder(h))

    def to_pose(self, z):                          # -> [B,T,15,3]
        return self.decode_raw(z).permute(0, 2, 1).reshape(-1, self.seq_len, 15, 3)

    def vae_loss(self, pose, kl_weight):
        mu, logvar = self.moments(pose)
        z = torch.randn_like(mu) * torch.exp(0.5 * logvar) + mu
        rec = self.decode_raw(z).permute(0, 2, 1)
        kld = torch.mean(-0.5 * torch.sum(1 + logvar - mu ** 2 - logvar.exp(), dim=1), dim=0)
        return F.mse_loss(rec, pose, reduction="sum") + kl_weight * kld


def fit_vae(shape, windows, steps=2000, batch=128, lr=2e-3, kl_weight=0.01, seed=0, device=None, latent_gain=1.0):
    """Adam on `vae_loss` over synthetic windows [n,T,45]; returns (CPU state_dict, reconstruction error in m).

    kl_weight: the reference trained with 0.5 on real motion; on the low-dimensional synthetic motion that
    collapses the posterior (reconstruction = the mean pose), 0.01 reconstructs to about 1 cm."""
    device = torch.device(device or ("cuda" if torch.cuda.is_available() else "cpu"))
    torch.manual_seed(seed)
    net = MotionVAE(shape.latent_dim, shape.seq_len, tuple(shape.hidden), shape.channels).to(device)
    data = torch.as_tensor(np.asarray(windows), dtype=torch.float32, device=device)
    with torch.no_grad():      # start from "output = mean pose": the network only has to learn the motion
        net.final_layer[3].bias.copy_(data.mean(dim=(0, 1)))
        net.final_layer[3].weight.mul_(0.1)
    g = torch.Generator(device="cpu").manual_seed(seed)
    opt = torch.optim.Adam(net.parameters(), lr=lr)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=lr, total_steps=steps)
    net.train()
    for _ in range(steps):
        idx = torch.randint(0, data.shape[0], (batch,), generator=g).to(device)
        opt.zero_grad(set_to_none=True)
        net.vae_loss(data[idx], kl_weight).backward()
        opt.step()
        sched.step()
    net.eval()
    with torch.no_grad():
        x = data[:256]
        rec = net.decode_raw(net.moments(x)[0]).permute(0, 2, 1)
        err = (rec - x).reshape(-1, 15, 3).norm(dim=-1).mean().item()
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    if latent_gain != 1.0:
        # change of latent gauge u = z / gain (same encoder->decoder function, prior N(0, 1/gain^2)): the decoder
        # becomes `gain` times more sensitive to the latent, as with a VAE trained under a stronger KL term
        g = float(latent_gain)
        sd["decoder_input.weight"] = sd["decoder_input.weight"] * g
        sd["fc_mu.weight"], sd["fc_mu.bias"] = sd["fc_mu.weight"] / g, sd["fc_mu.bias"] / g
        sd["fc_var.bias"] = sd["fc_var.bias"] - 2.0 * float(np.log(g))
    return sd, err


def _cli():
    """Fit the two motion VAEs on synthetic motion and write them where the reference looks for its checkpoints
    (`optimizer.py:334,344`, schema of `networks/train.py:102-108`), so that `main()` / `whole_sequence` run end to end
    without the external weight download.  Accuracy experiments only: real weights come from the authors."""
    import argparse
    import os
    from . import synth, vae as V
    from .optimizer import GLOBAL_VAE_PATH, LOCAL_VAE_PATH
    p = argparse.ArgumentParser(description=_cli.__doc__)
    p.add_argument("--root", default=".", help="directory the relative checkpoint paths are resolved against")
    p.add_argument("--steps", type=int, default=2000)
    p.add_argument("--latent_dim", type=int, default=2048)
    p.add_argument("--seed", type=int, default=101)
    a = p.parse_args()
    shape = V.VAEShape(latent_dim=a.latent_dim)
    for path, relative, seed in ((LOCAL_VAE_PATH, False, a.seed), (GLOBAL_VAE_PATH, True, a.seed + 1)):
        windows = synth.make_training_windows(4096, shape.seq_len, seed)
        if relative:        # relative-global poses drift with the camera: 4 mm / frame along x (synth.make_sequence)
            windows = windows.reshape(-1, shape.seq_len, 15, 3).copy()
            windows[..., 0] += (0.004 * np.arange(shape.seq_len))[None, :, None]
            windows = windows.reshape(-1, shape.seq_len, 45)
        sd, err = fit_vae(shape, windows, steps=a.steps, seed=seed, latent_gain=8.0)
        out = os.path.join(a.root, path)
        os.makedirs(os.path.dirname(out), exist_ok=True)
        V.save_checkpoint(out, sd)
        print("%s: reconstruction %.1f mm -> %s" % ("global" if relative else "local", err * 1e3, out))


if __name__ == "__main__":
    _cli()
