"""Skeleton constants shared by the window optimiser, the metrics and the synthetic data.

The 15-joint egocentric skeleton and its kinematic tree are fixed by the reference
(`optimizer.py:34`, `utils/skeleton.py:17-22`); everything on the hot path indexes joints in
this order.
"""
import numpy as np

N_JOINTS = 15
JOINT_NAMES = (
    "Neck", "Right_shoulder", "Right_elbow", "Right_wrist", "Left_shoulder", "Left_elbow",
    "Left_wrist", "Right_hip", "Right_knee", "Right_ankle", "Right_foot", "Left_hip",
    "Left_knee", "Left_ankle", "Left_foot",
)
# parent of joint j; joint 0 is its own parent (zero-length "bone", optimizer.py:34)
KINEMATIC_PARENTS = (0, 0, 1, 2, 0, 4, 5, 1, 7, 8, 9, 4, 11, 12, 13)

# Mean skeleton in millimetres, 3 x 15 (x, y, z rows). These are the values of the reference's
# data file utils/fisheye/mean3D.mat (key 'mean3D'), which calculate_errors.py:149-156 uses for the
# bone-length-normalised MPJPE. Data, not code.
MEAN3D_MM = np.array([
    [6.12454847, 145.97761, 258.72083056, 281.27554815, -130.58758154,
     -217.63663461, -234.47818229, 122.57391072, 157.99031993, 172.09879492,
     215.33356937, -52.15750419, -59.0959752, -36.18717374, -80.10264932],
    [233.90813433, 232.60823975, 188.18493809, 72.79136312, 239.16565076,
     203.68825151, 91.05888921, 239.95855861, 133.01398165, 176.20098748,
     37.42165039, 243.04617535, 149.38252591, 180.44482382, 44.79721165],
    [176.25176082, 220.73112637, 404.39836013, 488.37987609, 232.02432922,
     436.14841643, 529.22255096, 675.05067301, 1019.17833662, 1331.949378,
     1391.75072893, 683.67509016, 1037.58363271, 1353.00767289, 1407.87463384],
], dtype=np.float64)


def mean_bone_length_mm():
    """Bone lengths (mm) of the mean skeleton, entry 0 == 0 (utils/skeleton.py:102-110)."""
    m = MEAN3D_MM.T
    return np.linalg.norm(m - m[list(KINEMATIC_PARENTS)], axis=1)
