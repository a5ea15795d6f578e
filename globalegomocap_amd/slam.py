"""SLAM trajectory -> the scaled camera-to-world matrices of `camera_pose_list` (host, numpy float64).

Mirror of the reference's data preparation `MakeDataForOptimization/slam_reader.py:50-121` (the pickle's
`camera_pose_list`): trajectory lines `time tx ty tz qx qy qz qw`, frames selected by
`round(time * fps)`, poses made relative to the first selected frame, translation scaled by the similarity (Umeyama)
scale that maps the SLAM head trajectory onto the ground-truth head trajectory.  open3d's `PointCloud.transform` of
the reference is the plain `R p + t`.  Not on the timed path; pinned by `tests/golden/slam.npz`, which comes from the
reference's own `SLAMReader.read_trajectory` (`oracle/make_golden_slam.py`).
"""
import numpy as np
from scipy.spatial.transform import Rotation

from .errors import umeyama


def pose_matrix(trans, quat):
    """4x4 from translation + xyzw quaternion (slam_reader.py:16-25)."""
    m = np.eye(4)
    m[:3, :3] = Rotation.from_quat(np.asarray(quat, dtype=np.float64)).as_matrix()
    m[:3, 3] = np.asarray(trans, dtype=np.float64)
    return m


def parse_trajectory(lines, start_frame, end_frame, fps=30):
    """-> (trans [n,3], quat [n,4]) of the lines whose frame id round(t * fps) lies in [start_frame, end_frame)."""
    trans, quat = [], []
    for line in lines:
        f = line.strip().split()
        if not f:
            continue
        if start_frame <= round(float(f[0]) * fps) < end_frame:
            trans.append(np.array(f[1:4], dtype=np.float64))
            quat.append(np.array(f[4:], dtype=np.float64))
    return np.asarray(trans).reshape(-1, 3), np.asarray(quat).reshape(-1, 4)


def relative_poses(trans, quat):
    """Poses relative to the first one, returned as (trans, quat) like the reference (matrix -> quaternion -> matrix
    round trip included, slam_reader.py:153-166)."""
    m0_inv = np.linalg.inv(pose_matrix(trans[0], quat[0]))
    rt, rq = [], []
    for t, q in zip(trans, quat):
        m = m0_inv.dot(pose_matrix(t, q))
        rt.append(m[:3, 3].copy())
        rq.append(Rotation.from_matrix(m[:3, :3]).as_quat())
    return np.asarray(rt), np.asarray(rq)


def scaled_trajectory(trans, quat, scale=1.0):
    """read_trajectory (slam_reader.py:168-199): relative poses with the translation multiplied by `scale`."""
    rt, rq = relative_poses(trans, quat)
    return [pose_matrix(t * scale, q) for t, q in zip(rt, rq)]


def camera_pose_list(lines, local_pose_list, gt_global_pose, start_frame, end_frame, fps=30):
    """read_trajectory_new (slam_reader.py:50-121) -> (list of 4x4, R_1, t_1): the scale is the Umeyama scale between the
    head joint (index 0) carried along the un-scaled SLAM poses and the ground-truth head positions."""
    trans, quat = parse_trajectory(lines, start_frame, end_frame, fps)
    rt, rq = relative_poses(trans, quat)
    gt = np.asarray(gt_global_pose, dtype=np.float64)
    slam_head = np.stack([pose_matrix(t, q)[:3, :3] @ np.asarray(local_pose_list[i], dtype=np.float64)[0] + t
                          for i, (t, q) in enumerate(zip(rt, rq))])
    gt_head = gt[:len(rt), 0]
    c, _, _ = umeyama(slam_head, gt_head)
    _, R_1, t_1 = umeyama(gt_head, slam_head)
    return [pose_matrix(t * c, q) for t, q in zip(rt, rq)], R_1, t_1
