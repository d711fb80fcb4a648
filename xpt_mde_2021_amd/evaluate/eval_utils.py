"""Depth evaluation with the reference's names (evaluate/eval_utils.py:109-154): valid_depth_filter (Garg crop,
median scaling, clipping) and compute_depth_metrics -- plain numpy, the step right after the hot path."""
import numpy as np

from ..config import opts


def valid_depth_filter(depth_pred, depth_true):
    """eval_utils.py:109-131.  depth_pred, depth_true [height, width] -> (pred [N], true [N])."""
    depth_pred = np.squeeze(depth_pred).astype(np.float64).copy()
    depth_true = np.squeeze(depth_true)
    mask = np.logical_and(depth_true > opts.MIN_DEPTH, depth_true < opts.MAX_DEPTH)
    gt_height, gt_width = depth_true.shape
    crop = np.array([0.40810811 * gt_height, 0.99189189 * gt_height,
                     0.03594771 * gt_width, 0.96405229 * gt_width]).astype(np.int32)
    crop_mask = np.zeros(mask.shape, dtype=bool)
    crop_mask[crop[0]:crop[1], crop[2]:crop[3]] = True
    mask = np.logical_and(mask, crop_mask)
    scaler = np.median(depth_true[mask]) / np.median(depth_pred[mask])
    depth_pred[mask] *= scaler
    depth_pred = np.clip(depth_pred, opts.MIN_DEPTH, opts.MAX_DEPTH)
    return depth_pred[mask], depth_true[mask]


def compute_depth_metrics(pred, gt):
    """eval_utils.py:134-154 -> [abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3]."""
    thresh = np.maximum(gt / pred, pred / gt)
    a1, a2, a3 = (thresh < 1.25).mean(), (thresh < 1.25 ** 2).mean(), (thresh < 1.25 ** 3).mean()
    rmse = np.sqrt(((gt - pred) ** 2).mean())
    rmse_log = np.sqrt(((np.log(gt) - np.log(pred)) ** 2).mean())
    abs_rel = np.mean(np.abs(gt - pred) / gt)
    sq_rel = np.mean(((gt - pred) ** 2) / gt)
    return [abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3]


def evaluate_depth(depth_pred_batch, depth_true_batch):
    """Mean of the seven metrics over the frames of a prediction npz (evaluate_main.py:60-75)."""
    rows = []
    for pred, true in zip(depth_pred_batch, depth_true_batch):
        if (np.squeeze(true) > opts.MIN_DEPTH).sum() == 0:
            continue
        p, t = valid_depth_filter(pred, true)
        if p.size:
            rows.append(compute_depth_metrics(p, t))
    return np.mean(np.asarray(rows), axis=0) if rows else np.full(7, np.nan)


# ------------------------------------------------------------------------------------------------ pose metrics
def pose_rvec2matr_batch_np(poses):
    """Twist (tx, ty, tz, rx, ry, rz) [batch, N, 6] -> [batch, N, 4, 4] with the reference's negated-skew Rodrigues
    convention (utils/convert_pose.py:74-117): R = I + sin(t) W + (1 - cos(t)) W^2, W = -[w]_x."""
    poses = np.asarray(poses, dtype=np.float64)
    trans, uvec = poses[..., :3], poses[..., 3:]
    theta = np.linalg.norm(uvec, axis=-1, keepdims=True)
    safe = np.where(np.isclose(theta, 0), 1.0, theta)
    w = uvec / safe
    z = np.zeros_like(theta)
    w1, w2, w3 = w[..., 0:1], w[..., 1:2], w[..., 2:3]
    w_hat = np.concatenate([z, w3, -w2, -w3, z, w1, w2, -w1, z], axis=-1).reshape(poses.shape[:-1] + (3, 3))
    th = theta[..., None]
    rot = np.eye(3) + np.sin(th) * w_hat + (1.0 - np.cos(th)) * (w_hat @ w_hat)
    mat = np.zeros(poses.shape[:-1] + (4, 4))
    mat[..., :3, :3] = rot
    mat[..., :3, 3] = trans
    mat[..., 3, 3] = 1.0
    return mat


class PoseMetricNumpy:
    """Snippet pose errors (evaluate/eval_utils.py:9-87): absolute / scale-aligned trajectory error in metres and
    rotational error in radians, all poses re-based on the first frame of the snippet."""

    def __init__(self):
        self.trj_abs_err = np.array([])
        self.trj_rel_err = np.array([])
        self.rot_err = np.array([])

    def compute_pose_errors(self, pose_pred, pose_true_mat):
        """pose_pred [batch, numsrc, 6] twists, pose_true_mat [batch, numsrc, 4, 4] -> errors [batch, numsrc]."""
        if hasattr(pose_pred, "detach"):
            pose_pred = pose_pred.detach().cpu().numpy()
        if hasattr(pose_true_mat, "detach"):
            pose_true_mat = pose_true_mat.detach().cpu().numpy()
        pred = self.snippet_pose_from_first(pose_rvec2matr_batch_np(pose_pred))
        true = self.snippet_pose_from_first(np.asarray(pose_true_mat, dtype=np.float64))
        self.trj_abs_err = self.calc_trajectory_error(pred, true, True)
        self.trj_rel_err = self.calc_trajectory_error(pred, true, False)
        self.rot_err = self.calc_rotational_error(pred, true)

    def snippet_pose_from_first(self, poses):
        """[batch, numsrc, 4, 4] -> [batch, numsrc + 1, 4, 4]: the (identity) target pose inserted in the middle of the
        snippet, everything expressed relative to the first frame (eval_utils.py:27-40)."""
        target = np.tile(np.identity(4).reshape(1, 1, 4, 4), (poses.shape[0], 1, 1, 1))
        mats = np.concatenate([poses[:, :2], target, poses[:, 2:]], axis=1)
        return np.matmul(np.linalg.inv(mats[:, 0:1]), mats)

    def calc_trajectory_error(self, pose_pred_mat, pose_true_mat, abs_scale=False):
        xyz_pred, xyz_true = pose_pred_mat[:, :, :3, 3], pose_true_mat[:, :, :3, 3]
        if abs_scale:
            err = xyz_true - xyz_pred
        else:      # least-squares scale per frame: the monocular estimate has no absolute scale (0/0 at the origin frame)
            with np.errstate(invalid="ignore", divide="ignore"):
                scale = np.sum(xyz_true * xyz_pred, axis=2) / np.sum(xyz_pred ** 2, axis=2)
            err = xyz_true - xyz_pred * scale[..., np.newaxis]
        return np.sqrt(np.sum(err ** 2, axis=2))[:, 1:]

    def calc_rotational_error(self, pose_pred_mat, pose_true_mat):
        rel = np.matmul(np.linalg.inv(pose_pred_mat[:, :, :3, :3]), pose_true_mat[:, :, :3, :3])
        cosine = np.clip((np.trace(rel, axis1=2, axis2=3) - 1.0) / 2.0, -1.0, 1.0)
        return np.arccos(cosine)[:, 1:]

    def get_mean_pose_error(self):
        return np.mean(self.trj_abs_err), np.mean(self.trj_rel_err), np.mean(self.rot_err)
