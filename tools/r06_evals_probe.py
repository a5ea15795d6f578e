"""Developer probe (round 6): what makes a window leave L-BFGS early?  Evaluations per stage against the data's motion amplitude,
estimator noise and camera noise (bench.py's fitted synthetic VAEs, 240 windows per setting)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from globalegomocap_amd import synth, vae as V
from globalegomocap_amd.camera import FisheyeCamera, DEFAULT_CALIBRATION
from globalegomocap_amd.engine import WindowEngine, energy_weights, stats_to_numpy
dev = torch.device("cuda")
shape, cam = V.VAEShape(), FisheyeCamera.from_json(DEFAULT_CALIBRATION)
sd_l, _ = bench.fit_weights(shape, 101, dev, 2000, False)
sd_g, _ = bench.fit_weights(shape, 102, dev, 2000, True)
eng = WindowEngine(shape, cam, max_windows=256)
eng.load_vae(0, sd_l); eng.load_vae(1, sd_g); eng.set_precision("bf16")
wl, wg = energy_weights(0.01 / 10000, 0.001 / 100, 0.01, 0.0, 0.01), energy_weights(0.01, 0.001, 0.01, 0.0, 0.0)
n = 8 * 239 + 10
for level in (1.0, 0.3, 0.1, 0.02, 0.0):
    for jit in (bench.CAM_JITTER, None):
        act = np.zeros(n)  # quiet: blank heat-maps
        seq = synth.make_sequence(n, 77, cam, with_heatmaps=False, cam_jitter=jit, activity=act, quiet_level=level)
        d = synth.make_stream_device(n, 77, dev, camera=cam, host=seq)
        f0 = torch.as_tensor((8 * np.arange(240)).astype(np.int32), device=dev)
        mb = eng.mean_bone_length(d["est_local"]).reshape(1, 15).expand(240, 15).contiguous()
        eps = torch.randn(240, 2, shape.latent_dim, generator=torch.Generator().manual_seed(1))
        _, glob, stats = eng.optimize_windows(d["est_local"], d["cams"], d["heat"], f0, mb, eps[:, 0].contiguous().to(dev), eps[:, 1].contiguous().to(dev), wl, wg)
        st = stats_to_numpy(stats)
        ev = st["func_evals"].reshape(2, 240)
        it = st["n_iter"].reshape(2, 240)
        print("level %.2f cam jitter %-14s local evals mean %.1f (min %d max %d) iters %.1f | global evals mean %.1f (min %d, p10 %d, p90 %d, max %d) iters %.1f"
              % (level, jit, ev[0].mean(), ev[0].min(), ev[0].max(), it[0].mean(), ev[1].mean(), ev[1].min(), np.percentile(ev[1], 10), np.percentile(ev[1], 90), ev[1].max(), it[1].mean()))
