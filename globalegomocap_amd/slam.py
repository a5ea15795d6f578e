"""SLAM trajectory -> the scaled camera-to-world matrices of `camera_pose_list` (host, numpy float64).

Mirror of the reference's data preparation `MakeDataForOptimization/slam_reader.py:50-121` (the pickle's
`camera_pose_list`): trajectory lines `time tx ty tz qx qy qz qw`, frames selected by
`round(time * fps)`, poses made relative to the first selected frame, translation scaled by the similarity (Umeyama)
scale that maps the SLAM head trajectory onto the ground-truth head trajectory.  open3d's `PointCloud.transform` of
the reference is the plain `R p + t`.  Not on the timed path; pinned by `tests/golden/slam.npz`, which comes from the
reference's own `SLAMReader.read_trajectory` (`oracle/make_golden_slam.py`).

Everything is batched numpy / scipy (one text split, one `Rotation.from_quat` / `from_matrix` over all frames, matrix
products by `einsum`): no per-line or per-frame Python loop, so the 100 000-frame stream of BASELINE configs[4] converts in
a fraction of a second (`tools/slam_timing.py`).
"""
import numpy as np
from scipy.spatial.transform import Rotation

from .errors import umeyama


def pose_matrix(trans, quat):
    """4x4 from translation + xyzw quaternion (slam_reader.py:16-25); [n,3] + [n,4] -> [n,4,4]."""
    trans, quat = np.asarray(trans, dtype=np.float64), np.asarray(quat, dtype=np.float64)
    m = np.zeros(trans.shape[:-1] + (4, 4))
    m[..., :3, :3] = Rotation.from_quat(quat).as_matrix()
    m[..., :3, 3] = trans
    m[..., 3, 3] = 1.0
    return m


def parse_trajectory(lines, start_frame, end_frame, fps=30):
    """-> (trans [n,3], quat [n,4]) of the lines whose frame id round(t * fps) lies in [start_frame, end_frame).
    `lines`: an iterable of text lines `time tx ty tz qx qy qz qw` (blank lines are skipped) or the whole file as one string."""
    text = lines if isinstance(lines, str) else "\n".join(lines)
    rows = np.array(text.split(), dtype=np.float64).reshape(-1, 8)
    frame = np.rint(rows[:, 0] * fps)                   # python's round(): half to even, like rint
    keep = (frame >= start_frame) & (frame < end_frame)
    return rows[keep, 1:4].copy(), rows[keep, 4:8].copy()


def relative_poses(trans, quat):
    """Poses relative to the first one, returned as (trans, quat) like the reference (matrix -> quaternion -> matrix
    round trip included, slam_reader.py:153-166)."""
    m = pose_matrix(trans, quat)
    rel = np.einsum("ij,njk->nik", np.linalg.inv(m[0]), m)
    return rel[:, :3, 3].copy(), Rotation.from_matrix(rel[:, :3, :3]).as_quat()


def scaled_trajectory(trans, quat, scale=1.0):
    """read_trajectory (slam_reader.py:168-199): relative poses with the translation multiplied by `scale`; [n,4,4]."""
    rt, rq = relative_poses(trans, quat)
    return pose_matrix(rt * scale, rq)


def camera_pose_list(lines, local_pose_list, gt_global_pose, start_frame, end_frame, fps=30):
    """read_trajectory_new (slam_reader.py:50-121) -> ([n,4,4], R_1, t_1): the scale is the Umeyama scale between the
    head joint (index 0) carried along the un-scaled SLAM poses and the ground-truth head positions."""
    trans, quat = parse_trajectory(lines, start_frame, end_frame, fps)
    rt, rq = relative_poses(trans, quat)
    gt = np.asarray(gt_global_pose, dtype=np.float64)
    n = len(rt)
    head_local = np.asarray(local_pose_list, dtype=np.float64)[:n, 0]
    slam_head = np.einsum("nij,nj->ni", pose_matrix(rt, rq)[:, :3, :3], head_local) + rt
    gt_head = gt[:n, 0]
    c, _, _ = umeyama(slam_head, gt_head)
    _, R_1, t_1 = umeyama(gt_head, slam_head)
    return pose_matrix(rt * c, rq), R_1, t_1
