"""Developer tool: host enqueue time of one step vs its GPU time (is the path launch-bound?)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from globalegomocap_amd import synth, vae as V
from globalegomocap_amd.camera import FisheyeCamera, DEFAULT_CALIBRATION
from globalegomocap_amd.engine import WindowEngine, energy_weights
from globalegomocap_amd.sequence import window_starts
dev = torch.device("cuda"); shape = V.VAEShape(); cam = FisheyeCamera.from_json(DEFAULT_CALIBRATION)
sd = V.synthetic_state_dict(shape, 5)
seq = synth.make_sequence_device(2000, 1000, dev, cam)
starts = np.concatenate([c * 100 + window_starts(100) for c in range(20)]).astype(np.int32)
B = len(starts)
eng = WindowEngine(shape, cam, max_windows=B); eng.load_vae(0, sd); eng.load_vae(1, sd)
mb = eng.mean_bone_length(seq["est_local"][:100]).reshape(1, 15).expand(B, 15).contiguous()
eps = torch.randn(B, 2048, device=dev)
f0 = torch.as_tensor(starts, device=dev)
wl, wg = energy_weights(1e-1, 1e-1, 1.0, 0, 1e-2), energy_weights(1.0, 0.1, 1.0, 0, 0)
for mode in ("f32", "bf16"):
    eng.set_precision(mode)
    for _ in range(2):
        eng.optimize_windows(seq["est_local"], seq["cams"], seq["heat"], f0, mb, eps, eps, wl, wg, want_stats=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.optimize_windows(seq["est_local"], seq["cams"], seq["heat"], f0, mb, eps, eps, wl, wg, want_stats=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%s: host enqueue %.2f ms, until GPU done %.2f ms" % (mode, (t1 - t0) * 1e3, (t2 - t0) * 1e3))
