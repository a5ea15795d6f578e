"""Host time of the SLAM trajectory -> camera_pose_list conversion (globalegomocap_amd/slam.py) on a 100 000-pose stream,
the size BASELINE configs[4] names.   python tools/slam_timing.py [n_poses]"""
import os
import sys
import time

import numpy as np
from scipy.spatial.transform import Rotation

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from globalegomocap_amd import slam

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
rng = np.random.default_rng(0)
t = np.arange(n) / 30.0
trans = np.cumsum(rng.normal(0, 0.002, (n, 3)), axis=0)
quat = Rotation.from_rotvec(np.cumsum(rng.normal(0, 0.002, (n, 3)), axis=0)).as_quat()
text = "\n".join(" ".join("%.9f" % v for v in r) for r in np.concatenate([t[:, None], trans, quat], axis=1))
local = rng.normal(0, 0.2, (n, 15, 3))
gt = rng.normal(0, 1.0, (n, 15, 3))
t0 = time.perf_counter()
tr, q = slam.parse_trajectory(text, 0, n)
t1 = time.perf_counter()
mats = slam.scaled_trajectory(tr, q, 1.7)
t2 = time.perf_counter()
mats2, R1, t1_ = slam.camera_pose_list(text, local, gt, 0, n)
t3 = time.perf_counter()
print("%d poses: parse %.3f s, read_trajectory %.3f s, read_trajectory_new (parse + Umeyama scale) %.3f s"
      % (len(tr), t1 - t0, t2 - t1, t3 - t2))
