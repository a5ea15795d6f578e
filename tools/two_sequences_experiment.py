"""Developer experiment: two engines, each optimising a FULL 240-window sequence on its own HIP stream,
the second one started half a step late (its global stage overlaps the other's local stage)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from globalegomocap_amd import synth, vae as V
from globalegomocap_amd.camera import FisheyeCamera, DEFAULT_CALIBRATION
from globalegomocap_amd.engine import WindowEngine, energy_weights, LOCAL_STAGE
from globalegomocap_amd.sequence import window_starts

dev = torch.device("cuda")
shape = V.VAEShape(); cam = FisheyeCamera.from_json(DEFAULT_CALIBRATION)
cache = sys.argv[1] if len(sys.argv) > 1 else None
if cache and os.path.exists(cache):
    sd_l, _, sd_g, _ = torch.load(cache, weights_only=False)
else:
    sd_l, _ = bench.fit_weights(shape, 101, dev, 2000, False)
    sd_g, _ = bench.fit_weights(shape, 102, dev, 2000, True)
seq = synth.make_sequence_device(2000, 1000, dev, cam, cam_jitter=bench.CAM_JITTER)
starts = np.concatenate([c * 100 + window_starts(100) for c in range(20)]).astype(np.int32)
B = len(starts)
g = torch.Generator().manual_seed(4321)
eps = torch.randn(2 * B, 2048, generator=g).reshape(B, 2, -1)
wl, wg = energy_weights(1e-6, 1e-5, 0.01, 0, 0.01), energy_weights(0.01, 0.001, 0.01, 0, 0)

def make(n):
    parts = []
    for p in range(n):
        e = WindowEngine(shape, cam, max_windows=B)
        e.load_vae(0, sd_l); e.load_vae(1, sd_g)
        mb = e.mean_bone_length(seq["est_local"][:100]).reshape(1, 15).expand(B, 15).contiguous()
        parts.append(dict(e=e, f0=torch.as_tensor(starts, device=dev), mb=mb, el=eps[:, 0].contiguous().to(dev),
                          eg=eps[:, 1].contiguous().to(dev), s=torch.cuda.Stream()))
    return parts

def run(parts, steps, stagger):
    torch.cuda.synchronize(); t = time.perf_counter()
    if stagger and len(parts) > 1:
        p = parts[1]
        with torch.cuda.stream(p["s"]):       # half a step of extra work on stream 1 puts it out of phase
            x = seq["est_local"][(p["f0"].long()[:, None] + torch.arange(10, device=dev)[None])].contiguous()
            p["e"].optimize_stage(LOCAL_STAGE, x, p["mb"], p["el"], wl, heat=seq["heat"], frame0=p["f0"], want_stats=False)
    for _ in range(steps):
        for p in parts:
            with torch.cuda.stream(p["s"]):
                p["e"].optimize_windows(seq["est_local"], seq["cams"], seq["heat"], p["f0"], p["mb"], p["el"], p["eg"], wl, wg, want_stats=False)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) * 1e3

for n, stagger in ((1, False), (2, False), (2, True), (3, False)):
    parts = make(n)
    run(parts, 2, stagger)
    steps = 8
    ms = run(parts, steps, stagger)
    extra = 0.5 if stagger else 0.0
    per = ms / (steps * n + extra)
    print("engines %d stagger %d: %.2f ms per sequence  %.0f windows/s" % (n, stagger, per, B / per * 1e3), flush=True)
    del parts
