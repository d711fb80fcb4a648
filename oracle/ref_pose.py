"""oracle (test infrastructure): twist <-> matrix pose algebra.

Follows utils/convert_pose.py:32-71 (pose_rvec2matr_batch_tf) and
utils/convert_pose.py:151-168 (pose_matr2rvec_batch) of the reference.
"""
import torch


def pose_rvec2matr_batch(poses):
    """utils/convert_pose.py:32-71.

    poses: (tx, ty, tz, u1, u2, u3) [batch, N, 6] -> [batch, N, 4, 4].
    Rodrigues with the NEGATED skew matrix (convert_pose.py:56):
    R = I + sin(t) W + (1 - cos(t)) W W,  W = -[u]x,  t = |u|.
    where(|t| < 1e-8, I, R) as in :64.
    """
    poses = poses.unsqueeze(-1)                      # [B, N, 6, 1]  (:38)
    batch, snippet = poses.shape[:2]
    trans = poses[:, :, :3]                          # [B, N, 3, 1]
    uvec = poses[:, :, 3:]
    unorm = torch.linalg.vector_norm(uvec, dim=2, keepdim=True)   # [B, N, 1, 1] (:44)
    uvec = uvec / unorm                              # (:45) NaN at exactly zero rotation, as in the reference
    w1 = uvec[:, :, 0:1]
    w2 = uvec[:, :, 1:2]
    w3 = uvec[:, :, 2:3]
    z = torch.zeros((batch, snippet, 1, 1), dtype=poses.dtype)
    # sign-flipped skew matrix (:56)
    w_hat = torch.cat([z, w3, -w2, -w3, z, w1, w2, -w1, z], dim=2)
    w_hat = w_hat.reshape(batch, snippet, 3, 3)
    identity = torch.eye(3, dtype=poses.dtype).reshape(1, 1, 3, 3).expand(batch, snippet, 3, 3)
    tmpmat = identity + w_hat * torch.sin(unorm) + torch.matmul(w_hat, w_hat) * (1 - torch.cos(unorm))
    rotmat = torch.where(torch.abs(unorm) < 1e-8, identity, tmpmat)   # (:64)
    tmat = torch.cat([rotmat, trans], dim=3)         # [B, N, 3, 4]
    last_row = torch.tensor([0, 0, 0, 1], dtype=poses.dtype).reshape(1, 1, 1, 4).expand(batch, snippet, 1, 4)
    tmat = torch.cat([tmat, last_row], dim=2)
    return tmat.reshape(batch, snippet, 4, 4)


def pose_matr2rvec_batch(poses):
    """utils/convert_pose.py:151-168.  [batch, numsrc, 4, 4] -> [batch, numsrc, 6]."""
    R = poses[:, :, :3, :3]
    trace = R[:, :, 0, 0] + R[:, :, 1, 1] + R[:, :, 2, 2]
    theta = torch.acos((trace - 1.) / 2.)
    theta = theta.unsqueeze(-1)
    axis = torch.stack([R[:, :, 1, 2] - R[:, :, 2, 1],
                        R[:, :, 2, 0] - R[:, :, 0, 2],
                        R[:, :, 0, 1] - R[:, :, 1, 0]], dim=-1)
    rvec = torch.where(torch.abs(theta) < 0.00001, axis / 2., axis / (2 * torch.sin(theta)) * theta)
    trans = poses[:, :, :3, 3]
    return torch.cat([trans, rvec], dim=-1)
