"""Developer check: the fused compaction (gemm_rows.h) against compact_kernel -- run once per setting, compare the dumps bitwise.

    python tools/check_fused_compaction.py out_a.npz;  GEM_NO_FUSED_COMPACT=1 python tools/check_fused_compaction.py out_b.npz
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from globalegomocap_amd import synth, vae as V
from globalegomocap_amd.engine import WindowEngine, energy_weights
from globalegomocap_amd.camera import FisheyeCamera, DEFAULT_CALIBRATION
shape = V.VAEShape()
cam = FisheyeCamera.from_json(DEFAULT_CALIBRATION)
sd_l, sd_g = V.structured_state_dict(shape, 7), V.structured_state_dict(shape, 8, feature_offset=3.0)
out = {}
for B in (64, 240, 300):
    seq = synth.make_sequence(n_frames=8 * B + 10, seed=11, camera=cam, cam_jitter=(0.3, 0.002))
    est = np.asarray(seq["estimated_local_skeleton"], dtype=np.float32)
    heat = np.asarray(seq["heatmap_list"], dtype=np.float32)
    starts = (8 * np.arange(B)).astype(np.int32)
    pose = np.stack([est[s:s + 10] for s in starts])
    eps = np.random.default_rng(3).normal(size=(B, 2048)).astype(np.float32)
    e = WindowEngine(shape, cam, max_windows=B)
    e.load_vae(0, sd_l); e.load_vae(1, sd_g)
    mb = e.mean_bone_length(est)
    for st, w in ((0, (1e-6, 1e-5, 1e-2, 0.0, 1e-2)), (1, (1e-2, 1e-3, 1e-2, 0.0, 0.0))):
        o, s = e.optimize_stage(st, pose, mb, eps, energy_weights(*w), heat, starts)
        out["pose_%d_%d" % (B, st)] = o.cpu().numpy(); out["stats_%d_%d" % (B, st)] = s.cpu().numpy()
    e.close()
np.savez(sys.argv[1], **out)
print("wrote", sys.argv[1], {k: v.shape for k, v in out.items()})
