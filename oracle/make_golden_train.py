#!/usr/bin/env python3
"""Golden run of the UNMODIFIED reference's training step -> tests/golden/train_tiny.npz.

TEST INFRASTRUCTURE.  Runs only where /root/reference exists (never on the GPU box, never from tests).  Imports the reference's
`ConvVAE` (networks/models/SeqConvVAE.py) and drives it exactly as networks/train.py:77-83 does -- `network.train()`,
`optimizer.zero_grad()`, `network(batch)`, `network.loss_function(..., M_N=...)` (and, second case, `kl_weight=...`: the summed
form), `loss.backward()`, `torch.optim.Adam(lr, weight_decay).step()` -- on a small network whose widths are NOT multiples of 64
(so that the padding of the device layout is exercised), fp64 off (plain fp32 torch on the CPU, like the reference).

Inputs are regenerated from seeds by this repo (`vae_train.initial_state_dict`, `synth.make_training_windows`); the
reparameterisation noise is drawn here and handed to the reference through a spy on `torch.randn_like` (nothing inside the
reference is changed).  Stored per case: the noise, per step the three loss values, the gradients of step 0, and after the last
step the parameters, the running statistics and Adam's moments of the 1-D parameters.

    python oracle/make_golden_train.py        # a few seconds
    python oracle/make_golden_train.py --full # + tests/golden/train_full.npz: ONE step of the reference at its real size

`--full`: ConvVAE(latent_dim=2048, hidden [64,64,128,256,512], seq_len=10), batch 8, the `M_N` loss form of train.py:81 with
weight decay -- the 32.6 M-parameter layout (decoder_input / fc_mu|fc_var flattening, ConvTranspose1d taps) pinned to the
reference itself.  130 MB of gradients are not a fixture: stored are the three losses, every tensor's gradient 2-norm and
maximum, 256 sampled gradient entries per tensor (flat indices drawn from seed 99, stored), and the same sampled entries of the
parameters after the Adam step.
"""
import os
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))
sys.dont_write_bytecode = True

from globalegomocap_amd import synth, vae as vae_schema                  # noqa: E402
from globalegomocap_amd.vae_train import initial_state_dict              # noqa: E402
from make_golden import import_reference, to_torch_sd, OUT               # noqa: E402

CASES = {
    # name: shape, batch, steps, Adam lr / weight_decay, loss form and weight
    "mn": dict(shape=vae_schema.VAEShape(latent_dim=40, hidden=(24, 40)), batch=12, steps=4, lr=1e-3, wd=1e-3, form="M_N", w=0.02,
               init_seed=3, data_seed=5),
    "sum": dict(shape=vae_schema.VAEShape(latent_dim=24, hidden=(16, 32, 40)), batch=9, steps=3, lr=2e-3, wd=0.0, form="kl_weight", w=0.01,
                init_seed=4, data_seed=6),
}


def run_case(torch, ConvVAE, c):
    shape = c["shape"]
    init = initial_state_dict(shape, c["init_seed"])
    net = ConvVAE(in_channels=45, out_channels=45, latent_dim=shape.latent_dim, seq_len=shape.seq_len, hidden_dims=list(shape.hidden))
    net.load_state_dict(to_torch_sd(torch, init), strict=False)          # (num_batches_tracked is not in the float schema)
    opt = torch.optim.Adam(params=net.parameters(), lr=c["lr"], weight_decay=c["wd"])
    data = synth.make_training_windows(c["batch"] * c["steps"], shape.seq_len, c["data_seed"]).astype(np.float32)
    rng = np.random.default_rng(c["data_seed"] + 100)
    eps = rng.standard_normal((c["steps"], c["batch"], shape.latent_dim)).astype(np.float32)
    out = {"eps": eps, "poses": data.reshape(c["steps"], c["batch"], shape.seq_len, 45)}
    losses = []
    real_randn_like = torch.randn_like
    net.train()
    for s in range(c["steps"]):
        def spy(t, *a, _s=s, **k):
            assert tuple(t.shape) == eps[_s].shape
            return torch.from_numpy(eps[_s].copy())
        torch.randn_like = spy
        try:
            opt.zero_grad()
            batch = torch.from_numpy(out["poses"][s].copy())
            preds, inp, mu, log_var = net(batch)
            loss, rec, kld = net.loss_function(preds, inp, mu, log_var, **{c["form"]: c["w"]})
            loss.backward()
        finally:
            torch.randn_like = real_randn_like
        if s == 0:
            for k, p in net.named_parameters():
                out["grad0/" + k] = p.grad.detach().numpy().copy()
        opt.step()
        losses.append([loss.item(), rec.item(), kld.item()])
    out["losses"] = np.array(losses, np.float64)
    for k, v in net.state_dict().items():
        if not k.endswith("num_batches_tracked"):
            out["final/" + k] = v.detach().numpy().copy()
    st = opt.state_dict()["state"]
    for i, (k, p) in enumerate(net.named_parameters()):
        if p.dim() == 1:                 # (the matrices' moments are pinned through the parameters they produced)
            out["exp_avg/" + k] = st[i]["exp_avg"].numpy().copy()
            out["exp_avg_sq/" + k] = st[i]["exp_avg_sq"].numpy().copy()
    out["meta"] = np.array([c["batch"], c["steps"], shape.latent_dim, c["init_seed"], c["data_seed"]] + list(shape.hidden), np.int64)
    out["hyper"] = np.array([c["lr"], c["wd"], c["w"], 1.0 if c["form"] == "kl_weight" else 0.0], np.float64)
    out["init_sha256"] = np.array(vae_schema.state_dict_sha256(init, shape))
    return out


FULL_CASE = dict(shape=vae_schema.VAEShape(), batch=8, lr=1e-4, wd=1e-5, w=0.25 * 8 / 1000.0, init_seed=21, data_seed=22, n_sample=256)


def run_full(torch, ConvVAE):
    c = FULL_CASE
    shape = c["shape"]
    init = initial_state_dict(shape, c["init_seed"])
    net = ConvVAE(in_channels=45, out_channels=45, latent_dim=shape.latent_dim, seq_len=shape.seq_len, hidden_dims=list(shape.hidden))
    net.load_state_dict(to_torch_sd(torch, init), strict=False)
    opt = torch.optim.Adam(params=net.parameters(), lr=c["lr"], weight_decay=c["wd"])
    poses = synth.make_training_windows(c["batch"], shape.seq_len, c["data_seed"]).astype(np.float32)
    eps = np.random.default_rng(c["data_seed"] + 100).standard_normal((c["batch"], shape.latent_dim)).astype(np.float32)
    real_randn_like = torch.randn_like
    torch.randn_like = lambda t, *a, **k: torch.from_numpy(eps.copy())
    net.train()
    try:
        opt.zero_grad()
        preds, inp, mu, log_var = net(torch.from_numpy(poses.copy()))
        loss, rec, kld = net.loss_function(preds, inp, mu, log_var, M_N=c["w"])
        loss.backward()
    finally:
        torch.randn_like = real_randn_like
    out = {"eps": eps, "poses": poses, "losses": np.array([loss.item(), rec.item(), kld.item()], np.float64)}
    rng = np.random.default_rng(99)
    grads = {k: p.grad.detach().numpy().copy() for k, p in net.named_parameters()}
    opt.step()
    for k, p in net.named_parameters():
        g = grads[k].reshape(-1)
        idx = np.sort(rng.choice(g.size, size=min(c["n_sample"], g.size), replace=False)).astype(np.int64)
        out["idx/" + k] = idx
        out["grad/" + k] = g[idx]
        out["gnorm/" + k] = np.array([np.linalg.norm(g.astype(np.float64)), np.abs(g).max()], np.float64)
        out["param1/" + k] = p.detach().numpy().reshape(-1)[idx].copy()
    for k, v in net.state_dict().items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            out["final/" + k] = v.detach().numpy().copy()
    out["meta"] = np.array([c["batch"], shape.latent_dim, c["init_seed"], c["data_seed"]] + list(shape.hidden), np.int64)
    out["hyper"] = np.array([c["lr"], c["wd"], c["w"]], np.float64)
    out["init_sha256"] = np.array(vae_schema.state_dict_sha256(init, shape))
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    work = tempfile.mkdtemp(prefix="gem_golden_train_")
    torch, _, ConvVAE, _ = import_reference(work)
    torch.set_num_threads(1)
    if "--full" in sys.argv:
        blob = run_full(torch, ConvVAE)
        path = os.path.join(OUT, "train_full.npz")
        np.savez_compressed(path, **blob)
        print("  full size: losses %s" % blob["losses"])
        print("wrote %s (%.0f KB)" % (path, os.path.getsize(path) / 1024))
        return
    blob = {}
    for name, c in CASES.items():
        for k, v in run_case(torch, ConvVAE, c).items():
            blob[name + "/" + k] = v
        print("  case %s: losses %s" % (name, blob[name + "/losses"][:, 0]))
    path = os.path.join(OUT, "train_tiny.npz")
    np.savez_compressed(path, **blob)
    print("wrote %s (%.0f KB)" % (path, os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
