#!/usr/bin/env python3
"""Golden vectors of the input lifting step from the unmodified reference.

    python oracle/make_golden_lift.py        # needs /root/reference; writes tests/golden/lift.npz

The reference's `Skeleton.set_skeleton` (utils/skeleton.py:32-45: `get_max_preds` + `camera2world`) is imported and
run as is.  Two things of `set_skeleton_from_file` (utils/skeleton.py:74-90) cannot run in this image and are
done here in numpy, documented as such: `cv2` is not installed, so the INTER_NEAREST 64->1024 resize is a
`np.repeat` by 16 (identical for an integer ratio), and the .mat file reading is skipped (arrays are passed in).
`np.float` (removed from numpy 1.24) is re-created as an alias of `float` for `camera2world`.
Test infrastructure only.
"""
import os
import sys
import tempfile
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("GEM_REFERENCE", "/root/reference")
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)

from globalegomocap_amd import synth                                                      # noqa: E402
from globalegomocap_amd.camera import DEFAULT_CALIBRATION, ALT_CALIBRATION              # noqa: E402


def main():
    for name in ("open3d", "cv2", "natsort"):
        sys.modules.setdefault(name, types.ModuleType(name))
    if not hasattr(np, "float"):
        np.float = float
    sys.path.insert(0, REF)
    os.chdir(tempfile.mkdtemp(prefix="gem_golden_lift_"))
    from utils.skeleton import Skeleton

    rng = np.random.default_rng(21)
    F = 6
    centres = np.stack([rng.uniform(200, 1100, (F, 15)), rng.uniform(80, 950, (F, 15))], axis=-1)
    heat = synth.gaussian_heatmaps(centres[..., 0], centres[..., 1]).astype(np.float32)      # [F,64,64,15]
    heat += rng.uniform(0, 0.02, heat.shape).astype(np.float32)
    heat = heat.astype(np.float16).astype(np.float32)                                       # f16-exact: small fixture, many ties
    heat[1, :, :, 3] = 0.0                 # empty map: max == 0 -> masked to (0, 0)
    heat[2, :, :, 5] = -heat[2, :, :, 5] - 0.25     # all negative: the zero padding wins the argmax
    heat[3, :, :, 7] = 0.5                 # constant map: first texel wins
    heat[4, 63, 63, 0] = 4.0               # last texel
    heat[4, 0, 0, 1] = 4.0                 # first texel
    heat[5, 10, 20, 2] = 3.0; heat[5, 10, 21, 2] = 3.0; heat[5, 9, 40, 2] = 3.0     # ties: row-major first (9, 40)
    depth = rng.uniform(0.2, 1.8, (F, 15))
    out = {"heat": heat.astype(np.float16), "depth": depth}
    for tag, path in (("default", DEFAULT_CALIBRATION), ("alt", ALT_CALIBRATION)):
        sk = Skeleton(calibration_path=path)
        res = []
        for f in range(F):
            big = np.repeat(np.repeat(heat[f], 16, axis=0), 16, axis=1)                 # cv2.resize(..., INTER_NEAREST)
            big = np.pad(big, ((0, 0), (128, 128), (0, 0)), "constant", constant_values=0).transpose((2, 0, 1))
            res.append(np.array(sk.set_skeleton(big, depth[f], None, to_mesh=False)))
        out["skeleton_" + tag] = np.stack(res)
    np.savez_compressed(os.path.join(OUT, "lift.npz"), **out)
    print("wrote", os.path.join(OUT, "lift.npz"), {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
