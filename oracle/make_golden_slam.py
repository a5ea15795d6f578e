#!/usr/bin/env python3
"""Golden vectors of the SLAM-trajectory conversion from the unmodified reference.

    python oracle/make_golden_slam.py        # needs /root/reference; writes tests/golden/slam.npz

`SLAMReader.read_trajectory` (MakeDataForOptimization/slam_reader.py:168-199: frame selection, poses relative to the
first frame, scaled translation) is imported and run as is, with inert stand-ins for open3d / cv2 / natsort (never
touched on this call path) and `np.float` re-created as an alias of `float` (removed from numpy 1.24).
`read_trajectory_new` itself cannot run here (it transforms point clouds with open3d); its Umeyama scale is the
function already pinned through calculate_errors.  Test infrastructure only.
"""
import os
import sys
import tempfile
import types

import numpy as np
from scipy.spatial.transform import Rotation

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("GEM_REFERENCE", "/root/reference")
OUT = os.path.join(REPO, "tests", "golden")


def main():
    for name in ("open3d", "cv2", "natsort"):
        sys.modules.setdefault(name, types.ModuleType(name))
    if not hasattr(np, "float"):
        np.float = float
    sys.path.insert(0, os.path.join(REF, "MakeDataForOptimization"))
    work = tempfile.mkdtemp(prefix="gem_golden_slam_")
    os.chdir(work)
    from slam_reader import SLAMReader
    rng = np.random.default_rng(3)
    rows = []
    for i in range(40):
        t = i / 30.0 + rng.uniform(-0.004, 0.004)            # jittered time stamps: frame id = round(t * fps)
        q = Rotation.from_euler("xyz", [0.3 + 0.01 * i, -0.2 + 0.02 * i, 0.1 - 0.015 * i]).as_quat()
        p = np.array([1.0 + 0.1 * i, 0.02 * i * i, 0.3 - 0.05 * i]) + rng.normal(0, 0.01, 3)
        rows.append([t, *p, *q])
    rows = np.asarray(rows)
    with open("traj.txt", "w") as f:
        for r in rows:
            f.write(" ".join("%.9f" % v for v in r) + "\n")
    reader = SLAMReader(fps=30)
    out = {"rows": rows}
    for tag, (a, b, scale) in {"a": (0, 40, 1.0), "b": (5, 30, 2.37), "c": (12, 13, 0.5)}.items():
        out["range_" + tag] = np.array([a, b, scale])
        out["mats_" + tag] = np.asarray(reader.read_trajectory("traj.txt", a, b, scale=scale))
    np.savez_compressed(os.path.join(OUT, "slam.npz"), **out)
    print("wrote", os.path.join(OUT, "slam.npz"), {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
