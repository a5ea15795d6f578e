"""Developer experiment: does running two half-batches on two HIP streams beat one full batch?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from globalegomocap_amd import synth, vae as V
from globalegomocap_amd.camera import FisheyeCamera, DEFAULT_CALIBRATION
from globalegomocap_amd.engine import WindowEngine, energy_weights
from globalegomocap_amd.sequence import window_starts

dev = torch.device("cuda")
shape = V.VAEShape(); cam = FisheyeCamera.from_json(DEFAULT_CALIBRATION)
sd_l, _ = bench.fit_weights(shape, 101, dev, 2000, False)
sd_g, _ = bench.fit_weights(shape, 102, dev, 2000, True)
NC = int(os.environ.get("GEM_EXP_CHUNKS", "20"))        # 20 chunks = 240 windows; 128 = 1536; 683 = 8196
seq = synth.make_sequence_device(100 * NC, 1000, dev, cam, cam_jitter=bench.CAM_JITTER)
starts = np.concatenate([c * 100 + window_starts(100) for c in range(NC)]).astype(np.int32)
B = len(starts)
g = torch.Generator().manual_seed(4321)
eps = torch.randn(2 * B, 2048, generator=g).reshape(B, 2, -1)
wl, wg = energy_weights(1e-6, 1e-5, 0.01, 0, 0.01), energy_weights(0.01, 0.001, 0.01, 0, 0)

STAGGER = int(os.environ.get("GEM_EXP_STAGGER", "0"))      # shader cycles the k-th stream idles before its call (de-phases the lanes)


def make(n_parts):
    parts = []
    for p in range(n_parts):
        lo, hi = p * B // n_parts, (p + 1) * B // n_parts
        e = WindowEngine(shape, cam, max_windows=hi - lo)
        e.load_vae(0, sd_l); e.load_vae(1, sd_g)
        e.set_precision(os.environ.get("GEM_EXP_PRECISION", "f32"))
        if os.environ.get('GEM_EXP_GRAPHS'): e.enable_graphs(True)
        mb = e.mean_bone_length(seq["est_local"][:100]).reshape(1, 15).expand(hi - lo, 15).contiguous()
        parts.append(dict(e=e, f0=torch.as_tensor(starts[lo:hi], device=dev), mb=mb, el=eps[lo:hi, 0].contiguous().to(dev),
                          eg=eps[lo:hi, 1].contiguous().to(dev), s=torch.cuda.Stream()))
    return parts

def run(parts, steps):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(steps):
        for k, p in enumerate(parts):
            with torch.cuda.stream(p["s"]):
                if k and STAGGER:
                    torch.cuda._sleep(STAGGER * k)
                p["e"].optimize_windows(seq["est_local"], seq["cams"], seq["heat"], p["f0"], p["mb"], p["el"], p["eg"], wl, wg, want_stats=False)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / steps * 1e3

for n in (1, 2, 3):
    parts = None
    torch.cuda.empty_cache()
    parts = make(n)
    run(parts, 3)
    ms = run(parts, 8)
    print("streams %d: %.2f ms/step  %.0f windows/s" % (n, ms, B / ms * 1e3), flush=True)
