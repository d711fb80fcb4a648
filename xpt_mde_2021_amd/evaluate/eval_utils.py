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
