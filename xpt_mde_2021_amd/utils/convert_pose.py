"""Pose algebra of the hot path (reference: utils/convert_pose.py).

pose_rvec2matr_batch_tf keeps the reference's name (convert_pose.py:32-71) and runs the gfx950 kernel
(K0, fwd + bwd).  pose_matr2rvec_batch (convert_pose.py:151-168) is only ever applied to the constant
stereo extrinsic / its inverse ([B,1,4,4], no gradient), so it stays a few device-agnostic tensor ops.
"""
import torch

from ..hip import ops as _ops


def pose_rvec2matr_batch_tf(poses):
    """(tx, ty, tz, u1, u2, u3) [batch, N, 6] -> [batch, N, 4, 4]; negated-skew Rodrigues (:56)."""
    return _ops.pose_rvec2matr(poses)


pose_rvec2matr_batch = pose_rvec2matr_batch_tf


def pose_matr2rvec_batch(poses):
    """[batch, numsrc, 4, 4] -> twist [batch, numsrc, 6]  (convert_pose.py:151-168)."""
    R = poses[..., :3, :3]
    trace = R[..., 0, 0] + R[..., 1, 1] + R[..., 2, 2]
    theta = torch.acos((trace - 1.) / 2.).unsqueeze(-1)
    axis = torch.stack([R[..., 1, 2] - R[..., 2, 1], R[..., 2, 0] - R[..., 0, 2], R[..., 0, 1] - R[..., 1, 0]], dim=-1)
    rvec = torch.where(torch.abs(theta) < 0.00001, axis / 2., axis / (2 * torch.sin(theta)) * theta)
    return torch.cat([poses[..., :3, 3], rvec], dim=-1)


def rigid_inverse(T):
    """Inverse of rigid transforms [..., 4, 4] in closed form: [R t; 0 1]^-1 = [R^T, -R^T t; 0 1].  Stands in for the
    reference's tf.linalg.inv on poses (losses.py:90, 229): no LU factorisation, no host synchronisation, so it can be
    captured into the hipGraph step; identical to the general inverse up to rounding for valid poses."""
    R = T[..., :3, :3]
    t = T[..., :3, 3:]
    Rt = R.transpose(-1, -2)
    top = torch.cat([Rt, -(Rt @ t)], dim=-1)
    bottom = T[..., 3:, :]
    return torch.cat([top, bottom], dim=-2)
