#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the UNMODIFIED reference in the build container.

TEST INFRASTRUCTURE.  Runs only where /root/reference exists (never on the GPU box, never from
tests).  It imports the reference's own modules (`optimizer`, `networks.models.SeqConvVAE`,
`utils.fisheye.FishEyeCalibrated`, `calculate_errors`) after registering inert stand-ins for the
three visualisation / directory-sorting packages that are absent offline and never touched on the
path (open3d, cv2, natsort -- SURVEY.md section 8c), feeds them synthetic inputs built by this
repo's own `globalegomocap_amd.synth` / `.vae`, and stores inputs + the reference's outputs as
small fixtures.  Nothing from the reference tree is copied.

    python oracle/make_golden.py            # writes tests/golden/*.npz   (about 2 minutes)
"""
import os
import pickle
import sys
import tempfile
import types
from collections import OrderedDict

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("GEM_REFERENCE", "/root/reference")
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True

from globalegomocap_amd import synth, vae as vae_schema          # noqa: E402
from globalegomocap_amd.camera import FisheyeCamera, DEFAULT_CALIBRATION, ALT_CALIBRATION   # noqa: E402

TINY = vae_schema.VAEShape(latent_dim=32, hidden=(16, 16, 32, 32, 64))
FULL = vae_schema.VAEShape()
LOCAL_CKPT = "networks/logs/only_local_full_dataset_latent_2048_len_10_kl_0.5_2/checkpoints/19.pth.tar"
GLOBAL_CKPT = "networks/logs/real_full_dataset_latent_2048_len_10_slide_window_step_1_kl_0.5/checkpoints/19.pth.tar"


def import_reference(workdir):
    for name in ("open3d", "cv2", "natsort"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.path[:0] = [REF, os.path.join(REF, "networks")]
    os.symlink(os.path.join(REF, "utils"), os.path.join(workdir, "utils"))
    os.chdir(workdir)
    import torch
    import optimizer as ref_opt
    from models.SeqConvVAE import ConvVAE
    from utils.fisheye.FishEyeCalibrated import FishEyeCameraCalibrated
    return torch, ref_opt, ConvVAE, FishEyeCameraCalibrated


def to_torch_sd(torch, sd):
    return OrderedDict((k, torch.from_numpy(np.array(v))) for k, v in sd.items())


def np_sd(sd):
    return OrderedDict((k, v.detach().cpu().numpy().copy()) for k, v in sd.items())


def train_tiny(torch, ConvVAE, seed, steps=3000):
    """Briefly fit a tiny ConvVAE with the reference's own loss_function (SeqConvVAE.py:191-219)."""
    torch.manual_seed(seed)
    net = ConvVAE(in_channels=45, out_channels=45, latent_dim=TINY.latent_dim, seq_len=10, hidden_dims=list(TINY.hidden))
    data = torch.from_numpy(synth.make_training_windows(4096, 10, seed))
    opt = torch.optim.Adam(net.parameters(), lr=1e-2)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=1e-2, total_steps=steps)
    g = torch.Generator().manual_seed(seed)
    net.train()
    for it in range(steps):
        idx = torch.randint(0, data.shape[0], (64,), generator=g)
        out, inp, mu, logvar = net(data[idx])
        loss = net.loss_function(out, inp, mu, logvar, kl_weight=0.001)[0]
        opt.zero_grad(); loss.backward(); opt.step(); sched.step()
    net.eval()
    with torch.no_grad():
        rec = net(data[:256])[0]
        err = (rec - data[:256]).reshape(-1, 15, 3).norm(dim=-1).mean().item()
    print("  tiny VAE seed %d: recon error %.2f mm" % (seed, err * 1000))
    return np_sd(net.state_dict())


def main():
    os.makedirs(OUT, exist_ok=True)
    work = tempfile.mkdtemp(prefix="gem_golden_")
    torch, ref_opt, ConvVAE, FishEye = import_reference(work)
    torch.set_num_threads(1)          # fixed reduction order
    cam_json = DEFAULT_CALIBRATION
    cam = FisheyeCamera.from_json(cam_json)

    # ---------------------------------------------------------------- fisheye projection
    rng = np.random.default_rng(7)
    pts = synth.rest_skeleton()[None] + rng.normal(0, 0.15, size=(4, 15, 3))
    pts = pts.reshape(-1, 3)
    pts[0] = [1e-4, -2e-4, 0.5]            # near the optical axis
    pts[1] = [2.0, 0.1, 0.05]              # grazing, lands outside the 1280x1024 image
    pts[2] = [-0.3, 0.4, -0.2]             # behind the camera plane
    fe = {"points": pts.astype(np.float32)}
    for tag, path in (("default", DEFAULT_CALIBRATION), ("alt", ALT_CALIBRATION)):
        c = FishEye(path)
        fe["uv32_" + tag] = c.world2camera_pytorch(torch.from_numpy(pts.astype(np.float32))).numpy()
        fe["uv64_" + tag] = c.world2camera(pts.copy())
    np.savez_compressed(os.path.join(OUT, "fisheye.npz"), **fe)

    # ---------------------------------------------------------------- grid_sample as called at optimizer.py:147
    H = W = 64
    heat = rng.uniform(0, 1, size=(16, H, W)).astype(np.float16).astype(np.float32)    # f16-exact: small fixture
    uv = np.stack([rng.uniform(60, 1220, 16), rng.uniform(-60, 1090, 16)], axis=1).astype(np.float32)
    uv[0] = [128.0, 0.0]; uv[1] = [1152.0, 1024.0]; uv[2] = [128.0 + 1024 / 63 * 5, 1024 / 63 * 7]   # exact texels
    uvt = torch.from_numpy(uv).clone().requires_grad_(True)
    p2 = uvt.clone()
    p2 = torch.stack([p2[:, 0] - 128, p2[:, 1]], dim=1)
    p2 = ((p2 - 512) / 512).view(-1, 1, 1, 2)
    s = torch.nn.functional.grid_sample(torch.from_numpy(heat).view(-1, 1, H, W), p2, align_corners=True)
    s.sum().backward()
    np.savez_compressed(os.path.join(OUT, "grid_sample.npz"), heat=heat.astype(np.float16), uv=uv,
                        value=s.detach().numpy().reshape(-1), grad_uv=uvt.grad.numpy())

    # ---------------------------------------------------------------- operator level, tiny VAE, seeded weights
    sd = vae_schema.synthetic_state_dict(TINY, seed=11)
    net = ConvVAE(in_channels=45, out_channels=45, latent_dim=TINY.latent_dim, seq_len=10, hidden_dims=list(TINY.hidden))
    net.load_state_dict(to_torch_sd(torch, sd)); net.eval()
    seq = synth.make_sequence(n_frames=10, seed=3, camera=cam)
    pose = np.asarray(seq["estimated_local_skeleton"], dtype=np.float32)
    heat10 = np.asarray(seq["heatmap_list"], dtype=np.float32)
    ops = {"weights_seed": 11, "weights_checksum": vae_schema.state_dict_checksum(sd, TINY),
           "pose": pose, "heat_centres": seq["heatmap_centres"]}
    with torch.no_grad():
        mu, logvar = net.encode(torch.from_numpy(pose.reshape(1, 10, 45)).permute(0, 2, 1).contiguous())
    ops["mu"], ops["logvar"] = mu.numpy(), logvar.numpy()
    z = (mu + 0.3 * torch.from_numpy(rng.normal(size=(1, TINY.latent_dim)).astype(np.float32))).detach()
    ops["z"] = z.numpy()
    mean_skel = torch.from_numpy(pose).float()
    for tag, wts in (("local", dict(vae_weight=0.0, gmm_weight=0.0, smooth_weight=1e-5, bone_length_weight=1e-2,
                                     weight_3d=1e-6, reproj_weight=1e-2)),
                     ("global", dict(vae_weight=0.0, gmm_weight=0.0, smooth_weight=1e-3, bone_length_weight=1e-2,
                                      weight_3d=1e-2, reproj_weight=0)),
                     ("allterms", dict(vae_weight=3e-3, gmm_weight=0.0, smooth_weight=2e-2, bone_length_weight=5e-2,
                                        weight_3d=7e-3, reproj_weight=4e-2))):
        ck = os.path.join(work, "full_dummy.pth.tar")
        if not os.path.exists(ck):
            torch.save({"state_dict": to_torch_sd(torch, vae_schema.synthetic_state_dict(FULL, seed=5))}, ck)
        bpo = ref_opt.BodyPoseOptimizer(cam_json, mean_skel, ck, seq_len=10, network_seq_len=10, latent_dim=2048)
        bpo.network = net                       # swap in the tiny VAE; everything else is the reference
        bpo.set_weights(**wts)
        bpo.initial_pose = torch.from_numpy(pose).float()
        hs = torch.from_numpy(heat10).permute(0, 3, 1, 2).contiguous()
        bpo.heatmap_seq = hs.view(-1, 64, 64)
        zp = z.clone().requires_grad_(True)
        X = bpo.network.decode_to_bodypose(zp).squeeze(0).contiguous()
        parts = [bpo.pose_energy_3d(X), bpo.smooth_accelerate(X), bpo.bone_length_energy(X), bpo.vae_energy(X),
                 bpo.reprojection_energy_heatmap_fast(X)]
        total = bpo.total_loss(zp)
        total.backward()
        ops["X"] = X.detach().numpy()
        ops["parts_" + tag] = np.array([float(p.detach()) for p in parts])
        ops["total_" + tag] = float(total)
        ops["dz_" + tag] = zp.grad.numpy().copy()
        ops["mean_bone"] = bpo.mean_bone_length.numpy()
    np.savez_compressed(os.path.join(OUT, "ops_tiny.npz"), **ops)

    # ---------------------------------------------------------------- optimiser level
    def run_stage(bpo, pose_in, heat_in, seed):
        """One reference `optimize_pose_seq_pytorch_LBFGS` call with a closure trace."""
        trace, orig = [], bpo.total_loss

        def spy(zz):
            v = orig(zz)
            trace.append(float(v))
            return v
        bpo.total_loss = spy
        torch.manual_seed(seed)
        eps = torch.randn(1, bpo.network.latent_dim)         # what randn_like draws (SeqConvVAE.py:168)
        torch.manual_seed(seed)
        out = bpo.optimize_pose_seq_pytorch_LBFGS(pose_in, heat_in, pose_in.copy())
        bpo.total_loss = orig
        return out, np.array(trace), eps.numpy()[0]

    print("training tiny VAEs with the reference's loss ...")
    torch.set_num_threads(2)
    sd_local, sd_global = train_tiny(torch, ConvVAE, 21), train_tiny(torch, ConvVAE, 22)
    torch.set_num_threads(1)
    seqw = synth.make_sequence(n_frames=10, seed=1, camera=cam)
    posew = np.asarray(seqw["estimated_local_skeleton"])
    heatw = np.asarray(seqw["heatmap_list"])
    lb = {"pose": posew, "heat_centres": seqw["heatmap_centres"]}
    for k, v in sd_local.items():
        lb["local/" + k] = v
    for k, v in sd_global.items():
        lb["global/" + k] = v
    for tag, sdx, wts in (("local", sd_local, dict(vae_weight=0.0, gmm_weight=0.0, smooth_weight=1e-5,
                                                    bone_length_weight=1e-2, weight_3d=1e-6, reproj_weight=1e-2)),
                          ("global", sd_global, dict(vae_weight=0.0, gmm_weight=0.0, smooth_weight=1e-3,
                                                      bone_length_weight=1e-2, weight_3d=1e-2, reproj_weight=0)),
                          ("globalstrong", sd_global, dict(vae_weight=0.0, gmm_weight=0.0, smooth_weight=1e-1,
                                                            bone_length_weight=1e-2, weight_3d=1.0, reproj_weight=0))):
        net = ConvVAE(in_channels=45, out_channels=45, latent_dim=TINY.latent_dim, seq_len=10, hidden_dims=list(TINY.hidden))
        net.load_state_dict(to_torch_sd(torch, sdx)); net.eval()
        bpo = ref_opt.BodyPoseOptimizer(cam_json, torch.from_numpy(posew).float(), os.path.join(work, "full_dummy.pth.tar"),
                                        seq_len=10, network_seq_len=10, latent_dim=2048)
        bpo.network = net
        bpo.set_weights(**wts)
        out, trace, eps = run_stage(bpo, posew, heatw, 1234)
        lb[tag + "_out"], lb[tag + "_trace"], lb[tag + "_eps"] = out, trace, eps
        print("  tiny %s stage: %d evals, loss %.6g -> %.6g" % (tag, len(trace), trace[0], trace[-1]))
    np.savez_compressed(os.path.join(OUT, "lbfgs_tiny.npz"), **lb)

    # full-size network, seeded weights regenerated on both sides (only seed + checksum stored)
    sd_full = vae_schema.synthetic_state_dict(FULL, seed=5)
    bpo = ref_opt.BodyPoseOptimizer(cam_json, torch.from_numpy(posew).float(), os.path.join(work, "full_dummy.pth.tar"),
                                    seq_len=10, network_seq_len=10, latent_dim=2048)
    lf = {"weights_seed": 5, "weights_checksum": vae_schema.state_dict_checksum(sd_full, FULL), "pose": posew,
          "heat_centres": seqw["heatmap_centres"]}
    for tag, wts in (("local", dict(vae_weight=0.0, gmm_weight=0.0, smooth_weight=1e-5, bone_length_weight=1e-2,
                                     weight_3d=1e-6, reproj_weight=1e-2)),
                     ("global", dict(vae_weight=0.0, gmm_weight=0.0, smooth_weight=1e-3, bone_length_weight=1e-2,
                                      weight_3d=1e-2, reproj_weight=0)),
                     # random-init weights decode far from the input, the default weights then stop at
                     # the first test; stronger weights make the optimiser iterate (trace fixture)
                     ("localstrong", dict(vae_weight=1e-3, gmm_weight=0.0, smooth_weight=1e-1, bone_length_weight=1.0,
                                           weight_3d=1e-1, reproj_weight=1e-2)),
                     ("globalstrong", dict(vae_weight=0.0, gmm_weight=0.0, smooth_weight=1e-1, bone_length_weight=1.0,
                                            weight_3d=1.0, reproj_weight=0))):
        bpo.set_weights(**wts)
        out, trace, eps = run_stage(bpo, posew, heatw, 99)
        lf[tag + "_out"], lf[tag + "_trace"], lf[tag + "_eps"] = out, trace, eps
        with torch.no_grad():
            mu, logvar = bpo.network.encode(torch.from_numpy(posew.reshape(1, 10, 45)).float().permute(0, 2, 1).contiguous())
        z0 = mu + torch.from_numpy(eps)[None] * torch.exp(0.5 * logvar)
        zp = z0.clone().requires_grad_(True)
        bpo.initial_pose = torch.from_numpy(posew).float()
        bpo.heatmap_seq = torch.from_numpy(heatw).float().permute(0, 3, 1, 2).contiguous().view(-1, 64, 64)
        tot = bpo.total_loss(zp); tot.backward()
        lf[tag + "_z0"], lf[tag + "_loss0"], lf[tag + "_dz0"] = z0.numpy()[0], float(tot), zp.grad.numpy()[0]
        print("  full %s stage: %d evals, loss %.6g -> %.6g" % (tag, len(trace), trace[0], trace[-1]))
    np.savez_compressed(os.path.join(OUT, "lbfgs_full.npz"), **lf)

    # ---------------------------------------------------------------- pipeline level: the reference's main()
    RealConvVAE = ConvVAE

    def tiny_factory(**kw):
        # main() hard-codes latent_dim=2048 and default hidden dims (optimizer.py:332-350); build the tiny
        # network instead, and re-seed afterwards: module construction draws init weights from the global
        # RNG, so seeding here makes the eps stream start at window 0 (draw order: local, global, ...).
        net = RealConvVAE(**{**kw, "latent_dim": TINY.latent_dim, "hidden_dims": list(TINY.hidden)})
        torch.manual_seed(EPS_SEED)
        return net
    EPS_SEED = 1234
    ref_opt.ConvVAE = tiny_factory
    for rel, sdx in ((LOCAL_CKPT, sd_local), (GLOBAL_CKPT, sd_global)):
        os.makedirs(os.path.dirname(os.path.join(work, rel)), exist_ok=True)
        torch.save({"state_dict": to_torch_sd(torch, sdx)}, os.path.join(work, rel))
    seq100 = synth.make_sequence(n_frames=100, seed=2, camera=cam)
    os.makedirs(os.path.join(work, "data", "chunk0"))
    with open(os.path.join(work, "data", "chunk0", "test_data.pkl"), "wb") as f:
        pickle.dump({k: seq100[k] for k in ("estimated_local_skeleton", "gt_global_skeleton", "camera_pose_list",
                                              "heatmap_list")}, f)
    # the tiny latent (32-D) gives tiny latent gradients; with the CLI default weights the global stage
    # would stop at its first test (g.d > -1e-6).  weight_3d / smooth are CLI flags of the reference
    # (optimize_whole_sequence.py:15-18): run the pipeline at 100x so that both stages iterate.
    PIPE_W3D, PIPE_SMOOTH = 1.0, 0.1
    pl = {"seq_seed": 2, "eps_seed": 1234, "heat_centres": seq100["heatmap_centres"],
          "weight_3d": PIPE_W3D, "smooth": PIPE_SMOOTH}
    for tag, fs in (("smooth", True), ("raw", False)):
        torch.manual_seed(1234)
        errors, est_seq, mid_local, opt_seq, gt_seq = ref_opt.main(
            os.path.join("data", "chunk0"), camera_model_path=cam_json, vae_weight=0.0, gmm_weight=0.0,
            smoothness_weight=PIPE_SMOOTH, bone_length_weight=0.01, weight_3d=PIPE_W3D, reproj_weight=0.01, final_smooth=fs)
        pl["opt_" + tag] = np.asarray(opt_seq)
        pl["mid_local_" + tag] = np.asarray(mid_local)
        pl["est_" + tag] = np.asarray(est_seq)
        pl["gt_" + tag] = np.asarray(gt_seq)
        for k, v in errors.items():
            pl["err_%s/%s" % (tag, k)] = np.asarray(v)
        print("  main(final_smooth=%s): optimized_global_mpjpe %.3f mm (input %.3f mm)"
              % (fs, errors["optimized_global_mpjpe"] * 1000, errors["original_global_mpjpe"] * 1000))
    # self-noise of the reference: same run with 8 intra-op threads
    torch.set_num_threads(8)
    torch.manual_seed(1234)
    errors8 = ref_opt.main(os.path.join("data", "chunk0"), camera_model_path=cam_json, vae_weight=0.0, gmm_weight=0.0,
                           smoothness_weight=PIPE_SMOOTH, bone_length_weight=0.01, weight_3d=PIPE_W3D, reproj_weight=0.01,
                           final_smooth=True)[0]
    pl["err_smooth_8threads/optimized_global_mpjpe"] = np.asarray(errors8["optimized_global_mpjpe"])
    print("  reference self-noise 1 vs 8 threads: %.4f mm"
          % (abs(errors8["optimized_global_mpjpe"] - float(pl["err_smooth/optimized_global_mpjpe"])) * 1000))
    np.savez_compressed(os.path.join(OUT, "pipeline_tiny.npz"), **pl)
    for f in sorted(os.listdir(OUT)):
        print("%8.1f KB  %s" % (os.path.getsize(os.path.join(OUT, f)) / 1024, f))


if __name__ == "__main__":
    main()
