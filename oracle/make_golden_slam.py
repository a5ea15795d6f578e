#!/usr/bin/env python3
"""Golden vectors of the SLAM-trajectory conversion from the unmodified reference.

    python oracle/make_golden_slam.py        # needs /root/reference; writes tests/golden/slam.npz

`SLAMReader.read_trajectory` (MakeDataForOptimization/slam_reader.py:168-199: frame selection, poses relative to the
first frame, scaled translation) is imported and run as is, with inert stand-ins for open3d / cv2 / natsort (never
touched on this call path) and `np.float` re-created as an alias of `float` (removed from numpy 1.24).
`read_trajectory_new` (slam_reader.py:50-121) additionally carries the head joint along the SLAM poses with open3d's
`PointCloud.transform` before it estimates the similarity scale against the ground-truth head track.  open3d is absent
offline, so FOR THAT CALL ONLY this generator registers a two-line functional stand-in -- `PointCloud.transform(M)` as the
plain `R p + t` it is, `Vector3dVector` as `np.asarray` -- everything else (frame selection, relative poses, both Umeyama
fits, the scaled matrices) is the reference's own code.  Test infrastructure only.
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
    # ---- read_trajectory_new: scale from the head track (open3d's transform replaced by R p + t, see the module docstring)
    import open3d as o3d

    class _PointCloud:
        points = None

        def transform(self, M):
            M = np.asarray(M, dtype=np.float64)
            self.points = np.asarray(self.points, dtype=np.float64) @ M[:3, :3].T + M[:3, 3]
            return self
    o3d.geometry = types.SimpleNamespace(PointCloud=_PointCloud)
    o3d.utility = types.SimpleNamespace(Vector3dVector=lambda a: np.asarray(a, dtype=np.float64))
    sys.path.insert(0, REPO)
    from globalegomocap_amd import synth
    a, b = 5, 30
    n = b - a
    local = synth.make_motion(n, np.random.default_rng(11)) + rng.normal(0, 0.01, (n, 15, 3))      # local skeletons, camera frame
    # ground truth: the SLAM head track (relative poses applied to the head joint) under a similarity with scale 2.37, + noise
    sel = rows[a:b]
    m0 = np.eye(4); m0[:3, :3] = Rotation.from_quat(sel[0, 4:]).as_matrix(); m0[:3, 3] = sel[0, 1:4]
    heads = []
    for r in sel:
        m = np.eye(4); m[:3, :3] = Rotation.from_quat(r[4:]).as_matrix(); m[:3, 3] = r[1:4]
        rel = np.linalg.inv(m0) @ m
        heads.append(rel[:3, :3] @ local[len(heads), 0] + rel[:3, 3])
    heads = np.asarray(heads)
    Rg = Rotation.from_euler("zyx", [0.4, -0.3, 0.2]).as_matrix()
    gt = np.repeat((2.37 * heads @ Rg.T + np.array([0.5, -1.0, 2.0]))[:, None], 15, axis=1) + rng.normal(0, 0.005, (n, 15, 3))
    mats, R_1, t_1 = reader.read_trajectory_new("traj.txt", [p for p in local], [g for g in gt], a, b)
    out["new_range"] = np.array([a, b])
    out["new_local"], out["new_gt"] = local, gt
    out["new_mats"], out["new_R1"], out["new_t1"] = np.asarray(mats), np.asarray(R_1), np.asarray(t_1)
    print("read_trajectory_new: translation scale %.6f" % (np.linalg.norm(np.asarray(mats)[-1][:3, 3]) /
                                                          np.linalg.norm(np.asarray(out["mats_b"])[-1][:3, 3]) * 2.37))
    np.savez_compressed(os.path.join(OUT, "slam.npz"), **out)
    print("wrote", os.path.join(OUT, "slam.npz"), {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
