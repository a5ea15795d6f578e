"""Developer check (library built with -DGEM_TB_DEBUG_DUMP): the bf16 pose-gradient rows the energy terms leave, per row-tile variant."""
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
lib = _capi.load_library()
dump = torch.zeros(B * 10 * 64, dtype=torch.int16, device="cuda")
raw = C.CDLL(_capi.LIB_PATH)
assert raw.gem_debug_set_tb_dump(C.c_void_p(dump.data_ptr())) == 0
res = {}
for nrt in (2, 3):
    os.environ["GEM_TAIL16_NRT"] = str(nrt)
    eng = WindowEngine(FULL, FisheyeCamera.from_json(DEFAULT_CALIBRATION), max_windows=B)
    eng.load_vae(0, sd); eng.set_precision("bf16")
    dump.zero_()
    E, parts, dz, X = eng.energy_grad(0, z, pose, mb, energy_weights(1e-2, 1e-2, 1e-1, 1e-3, 1e-2), heat, starts)
    torch.cuda.synchronize()
    res[nrt] = (dump.cpu().numpy().reshape(B, 10, 64).copy(), dz.cpu().numpy())
    eng.close()
g2, g3 = res[2][0], res[3][0]
bad = np.argwhere(g2 != g3)
print("pose-gradient rows (bf16) differing entries:", len(bad), bad[:10].tolist())
for w, t, c in bad[:5]:
    print("  window %d t %d col %d (joint %d comp %d): %04x vs %04x" % (w, t, c, c // 3, c % 3, g2[w, t, c] & 0xFFFF, g3[w, t, c] & 0xFFFF))
print("dz differing windows:", np.unique(np.nonzero(res[2][1] != res[3][1])[0]))
