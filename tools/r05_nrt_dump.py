"""Developer check (library with gem_api.hip built -DGEM_DEBUG_EXPORTS; the tail's object file is the product's): the bf16 gradient rows
the tail hands to the backward front product, per row-tile variant."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from globalegomocap_amd import _capi
_capi.LIB_PATH = os.path.abspath(os.environ["GEM_LIB"])
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
raw = C.CDLL(_capi.LIB_PATH)
res = {}
for nrt in (2, 3):
    os.environ["GEM_TAIL16_NRT"] = str(nrt)
    eng = WindowEngine(FULL, FisheyeCamera.from_json(DEFAULT_CALIBRATION), max_windows=B)
    eng.load_vae(0, sd); eng.set_precision("bf16")
    E, parts, dz, X = eng.energy_grad(0, z, pose, mb, energy_weights(1e-2, 1e-2, 1e-1, 1e-3, 1e-2), heat, starts)
    torch.cuda.synchronize()
    bufs = []
    for which, cols in ((0, 256), (1, 256)):
        p = C.c_void_p()
        assert raw.gem_debug_buffer(eng._h, which, C.byref(p)) == 0
        t = torch.empty(B * 10 * cols, dtype=torch.int16, device="cuda")
        hip = C.CDLL("libamdhip64.so")
        assert hip.hipMemcpy(C.c_void_p(t.data_ptr()), p, C.c_size_t(t.numel() * 2), C.c_int(3)) == 0
        bufs.append(t.cpu().numpy().reshape(B, 10, cols).copy())
    res[nrt] = (bufs[0], bufs[1], dz.cpu().numpy())
    eng.close()
for name, i in (("g_out (gradient w.r.t. conv 0's pre-activation)", 0), ("a_in (conv 0's activation)", 1)):
    a, b = res[2][i], res[3][i]
    bad = np.argwhere(a != b)
    print(name, "differing entries:", len(bad), bad[:12].tolist())
    for w, t, c in bad[:6]:
        print("   window %d t %d col %d: %04x vs %04x" % (w, t, c, a[w, t, c] & 0xFFFF, b[w, t, c] & 0xFFFF))
print("dz differing windows:", np.unique(np.nonzero(res[2][2] != res[3][2])[0]))
