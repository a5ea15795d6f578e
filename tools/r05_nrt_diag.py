"""Developer check: which row-tile variant of the bf16 tail differs from which, on which output, by how much."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import __graft_entry__ as ge
if os.environ.get("GEM_LIB"):                  # a variant build of the library (A/B runs)
    from globalegomocap_amd import _capi
    _capi.LIB_PATH = os.path.abspath(os.environ["GEM_LIB"])
else:
    ge.build()
from globalegomocap_amd import synth, vae as V
from globalegomocap_amd.camera import FisheyeCamera, DEFAULT_CALIBRATION
from globalegomocap_amd.engine import WindowEngine, energy_weights
from oracle import np_oracle as O
FULL = V.VAEShape()
sd = V.structured_state_dict(FULL, 7, feature_offset=0.0)
B = 1100
seq = synth.make_sequence(n_frames=200, seed=41)
est = np.asarray(seq["estimated_local_skeleton"], dtype=np.float32)
heat = torch.as_tensor(np.asarray(seq["heatmap_list"], dtype=np.float32), device="cuda")
rng = np.random.default_rng(B)
starts = rng.integers(0, 190, B).astype(np.int32)
pose = np.stack([est[s:s + 10] for s in starts])
mb = O.mean_bone_length(est)
z = rng.normal(size=(B, FULL.latent_dim)).astype(np.float32) * 0.3
os.environ["GEM_DEV"] = "1"; os.environ["GEM_TAIL16"] = "1"
ROLL = int(os.environ.get("NRT_ROLL", "0"))
if ROLL:
    z, pose, starts = np.roll(z, ROLL, 0), np.roll(pose, ROLL, 0), np.roll(starts, ROLL, 0)
res = {}
for nrt in (1, 2, 3, 4, 5):
    os.environ["GEM_TAIL16_NRT"] = str(nrt)
    outs = []
    for rep in range(2):
        eng = WindowEngine(FULL, FisheyeCamera.from_json(DEFAULT_CALIBRATION), max_windows=B)
        eng.load_vae(0, sd); eng.set_precision("bf16")
        E, parts, dz, X = eng.energy_grad(0, z, pose, mb, energy_weights(*[float(v) for v in os.environ.get('NRT_W', '1e-2,1e-2,1e-1,1e-3,1e-2').split(',')]), heat, starts)
        torch.cuda.synchronize()
        outs.append((E.cpu().numpy(), parts.cpu().numpy(), dz.cpu().numpy(), X.cpu().numpy()))
        eng.close()
    print("nrt", nrt, "repeat identical:", [bool(np.array_equal(a, b)) for a, b in zip(*outs)])
    res[nrt] = outs[0]
for nrt in (1, 3, 4, 5):
    for name, a, b in zip(("E", "parts", "dz", "X"), res[nrt], res[2]):
        if not np.array_equal(a, b):
            bad = np.unique(np.nonzero(a != b)[0])
            print("nrt %d vs 2: %s differs on %d windows (first %s, mod G %s), max |d| %.3g rel %.3g" % (nrt, name, len(bad), bad[:8], bad[:8] % {1: 1, 3: 4, 4: 6, 5: 8}[nrt],
                  np.abs(a - b).max(), np.abs(a - b).max() / np.abs(b).max()))
            if name == "dz":
                k = int(bad[0]); idx = np.nonzero(a[k] != b[k])[0]
                print("   window %d: %d of %d dz entries differ; E %.9g, |dz| max %.3g; first entries %s: %s vs %s" % (k, len(idx), a.shape[1], res[2][0][k], np.abs(b[k]).max(), idx[:4], a[k][idx[:4]], b[k][idx[:4]]))

if os.environ.get("NRT_DUMP"):
    np.savez(os.environ["NRT_DUMP"], **{"n%d_%s" % (n, k): v for n, r in res.items() for k, v in zip(("E", "parts", "dz", "X"), r)})
