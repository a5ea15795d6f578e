"""Developer measurement: device time of the once-per-stage pieces (encode, final decode) at 240 windows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from globalegomocap_amd import vae as V
from globalegomocap_amd.camera import FisheyeCamera, DEFAULT_CALIBRATION
from globalegomocap_amd.engine import WindowEngine

B = 240
eng = WindowEngine(V.VAEShape(), FisheyeCamera.from_json(DEFAULT_CALIBRATION), max_windows=B)
eng.load_vae(0, V.synthetic_state_dict(V.VAEShape(), 1))
pose = torch.randn(B, 10, 45, device="cuda")
eps = torch.randn(B, 2048, device="cuda")
z = torch.randn(B, 2048, device="cuda")


def timeit(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


print("encode  %.1f us" % timeit(lambda: eng.encode(0, pose, eps)))
print("decode  %.1f us" % timeit(lambda: eng.decode(0, z)))
