"""Developer diagnostic: the 12 global stages of tests/golden/pipeline_full_nlglobal.npz (non-linear global VAE) on the HIP path,
with the composed front layer and with the two separate layers (GEM_NO_FRONT=1), against the reference's own counts / energies."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["GEM_DEV"] = "1"
import torch
import __graft_entry__ as ge
ge.build()
from helpers import full_golden_case, FULL
from globalegomocap_amd.camera import FisheyeCamera, DEFAULT_CALIBRATION
from globalegomocap_amd.engine import WindowEngine, energy_weights, stats_to_numpy
from globalegomocap_amd.sequence import window_starts

name = sys.argv[1] if len(sys.argv) > 1 else "pipeline_full_nlglobal"
g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
data, sd_l, sd_g, w_l, w_g = full_golden_case(g)
for tag in ("composed", "separate", "no_tail"):
    os.environ.pop("GEM_NO_FRONT", None); os.environ.pop("GEM_NO_TAIL", None)
    if tag == "separate": os.environ["GEM_NO_FRONT"] = "1"
    if tag == "no_tail": os.environ["GEM_NO_TAIL"] = "1"
    eng = WindowEngine(FULL, FisheyeCamera.from_json(DEFAULT_CALIBRATION), max_windows=12)
    eng.load_vae(0, sd_l); eng.load_vae(1, sd_g)
    mb = eng.mean_bone_length(data["estimated_local_skeleton"].astype(np.float32))
    rows = np.arange(1, 24, 2)
    out, stats = eng.optimize_stage(1, g["stage_in"][rows], mb, g["eps"][rows], energy_weights(*w_g), data["heatmap_list"], window_starts(100))
    tr = eng.read_trace(12)
    sn = stats_to_numpy(stats)
    out = out.cpu().numpy()
    print(tag)
    for k, row in enumerate(rows):
        ref = g["trace"][row]; n_ref = int(g["func_evals"][row])
        n = min(n_ref, int(sn["func_evals"][k]))
        rel = np.abs(tr[k, :n] - ref[:n]) / np.abs(ref[:n])
        first_bad = int(np.argmax(rel > 1e-4)) if (rel > 1e-4).any() else -1
        d = np.linalg.norm(out[k] - g["stage_out"][row], axis=-1)
        print("  row %2d evals %d/%d iters %d/%d loss %.7e/%.7e  pose diff %.4f (max %.4f) mm  trace rel max %.1e first>1e-4 at eval %d"
              % (row, sn["func_evals"][k], n_ref, sn["n_iter"][k], int(g["n_iter"][row]), sn["final_loss"][k], np.nanmin(ref), d.mean() * 1e3, d.max() * 1e3, rel.max(), first_bad))
    eng.close()
