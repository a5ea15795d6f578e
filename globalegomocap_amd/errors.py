"""Pose-error metrics of the sequence pipeline (host, numpy float64).

Mirror of the reference's `calculate_errors` (`calculate_errors.py:114-179`): same 18 keys, same
definitions.  This numpy version is the host CHECKER of the device implementation
(`gem_calculate_errors`, csrc/errors.hip, reached through `WindowEngine.calculate_errors` or
`main(..., device_metrics=True)`), and what `main()` uses when device metrics are off.
"""
from collections import OrderedDict

import numpy as np

from .skeleton import KINEMATIC_PARENTS, mean_bone_length_mm

_PARENTS = list(KINEMATIC_PARENTS)


def umeyama(P, Q):
    """Similarity Procrustes: (c, R, t) with Q ~ c * P @ R + t (utils/rigid_transform_with_scale.py:18-43)."""
    P, Q = np.asarray(P, dtype=np.float64), np.asarray(Q, dtype=np.float64)
    if P.shape != Q.shape:
        raise AssertionError("umeyama: shape mismatch")
    n = P.shape[0]
    mp, mq = P.mean(axis=0), Q.mean(axis=0)
    cov = (P - mp).T @ (Q - mq) / n
    V, S, W = np.linalg.svd(cov)
    if np.linalg.det(V) * np.linalg.det(W) < 0.0:
        S[-1] = -S[-1]
        V[:, -1] = -V[:, -1]
    R = V @ W
    c = S.sum() / P.var(axis=0).sum()
    return c, R, mq - mp @ (c * R)


def mpjpe(a, b):
    """Mean per-joint position error (calculate_errors.py:24-30)."""
    return float(np.mean(np.linalg.norm(np.asarray(a) - np.asarray(b), axis=-1)))


def root_error(a, b):
    """Error of the hip midpoint, the reference's "camera position" proxy (calculate_errors.py:33-47)."""
    a, b = np.asarray(a), np.asarray(b)
    ra, rb = (a[:, 7] + a[:, 11]) / 2, (b[:, 7] + b[:, 11]) / 2
    return float(np.mean(np.linalg.norm(ra - rb, axis=1)))


def align_sequence(est, gt):
    """One similarity transform for the whole sequence (calculate_errors.py:8-21)."""
    e = np.asarray(est, dtype=np.float64).reshape(-1, 3)
    g = np.asarray(gt, dtype=np.float64).reshape(-1, 3)
    c, R, t = umeyama(e, g)
    return (e @ R * c + t).reshape(-1, 15, 3)


def resize_skeleton(joints, bone_mm):
    """Re-grow the skeleton from joint 0 with fixed bone lengths (utils/skeleton.py:124-136)."""
    j = np.array(joints, dtype=np.float64)
    vec = j - j[_PARENTS]
    ln = np.linalg.norm(vec, axis=1)
    scale = np.concatenate(([0.0], bone_mm[1:] / ln[1:]))
    vec = vec * scale[:, None] / 1000.0
    for i in range(j.shape[0]):
        j[i] = j[_PARENTS[i]] + vec[i]
    return j


def align_frames(est, gt, normalise_bones=False):
    """Per-frame Procrustes, optionally after bone-length normalisation (calculate_errors.py:62-83)."""
    est = np.array(est, dtype=np.float64)
    gt = np.array(gt, dtype=np.float64)
    if normalise_bones:
        bl = mean_bone_length_mm()
        est = np.stack([resize_skeleton(p, bl) for p in est])
        gt = np.stack([resize_skeleton(p, bl) for p in gt])
    out = np.empty_like(est)
    for s in range(est.shape[0]):
        c, R, t = umeyama(est[s], gt[s])
        out[s] = est[s] @ R * c + t
    return out, gt


def calculate_errors(final_estimated_seq, mid_estimated_seq, final_optimized_seq, final_gt_seq):
    est, mid, opt, gt = (np.asarray(x, dtype=np.float64) for x in
                         (final_estimated_seq, mid_estimated_seq, final_optimized_seq, final_gt_seq))
    r = OrderedDict()
    r["original_global_mpjpe"] = mpjpe(est, gt)
    r["mid_global_mpjpe"] = mpjpe(mid, gt)
    r["optimized_global_mpjpe"] = mpjpe(opt, gt)
    r["original_camera_pos_error"] = root_error(est, gt)
    r["optimized_camera_pos_error"] = root_error(opt, gt)
    a_est, a_mid, a_opt = align_sequence(est, gt), align_sequence(mid, gt), align_sequence(opt, gt)
    r["original_aligned_camera_pos_error"] = root_error(a_est, gt)
    r["mid_aligned_camera_pose_error"] = root_error(a_mid, gt)
    r["optimized_aligned_camera_pos_error"] = root_error(a_opt, gt)
    r["original_aligned_global_mpjpe"] = mpjpe(a_est, gt)
    r["aligned_mid_seq_mpjpe"] = mpjpe(a_mid, gt)
    r["optimized_aligned_global_mpjpe"] = mpjpe(a_opt, gt)
    r["aligned_original_mpjpe"] = mpjpe(align_frames(est, gt)[0], gt)
    r["aligned_mid_optimized_mpjpe"] = mpjpe(align_frames(mid, gt)[0], gt)
    r["aligned_optimized_mpjpe"] = mpjpe(align_frames(opt, gt)[0], gt)
    # bone-length normalised: the reference re-normalises the (already normalised) GT on every call
    p_est, g1 = align_frames(est, gt, True)
    p_mid, g2 = align_frames(mid, g1, True)
    p_opt, g3 = align_frames(opt, g2, True)
    r["bone_length_aligned_original_mpjpe"] = mpjpe(p_est, g1)
    r["bone_length_aligned_mid_optimized_mpjpe"] = mpjpe(p_mid, g2)
    r["bone_length_aligned_optimized_mpjpe"] = mpjpe(p_opt, g3)
    r["joints_error"] = np.mean(np.linalg.norm(p_opt - g3, axis=2), axis=0)
    return r
