"""evaluate/eval_utils.py:109-154 restated checks: Garg crop + median scaling, the seven depth metrics."""
import numpy as np

from xpt_mde_2021_amd.config import opts
from xpt_mde_2021_amd.evaluate import eval_utils as eu


def test_metrics_perfect_and_scaled():
    gt = np.linspace(1, 50, 1000)
    m = eu.compute_depth_metrics(gt.copy(), gt)
    assert np.allclose(m[:4], 0) and np.allclose(m[4:], 1)
    m = eu.compute_depth_metrics(gt * 1.3, gt)             # 30 % over: abs_rel .3, a1 = 0, a2 = 1
    assert np.isclose(m[0], 0.3) and m[4] == 0 and m[5] == 1 and m[6] == 1
    assert np.isclose(m[3], np.log(1.3))
    assert np.isclose(m[2], 0.3 * np.sqrt((gt ** 2).mean()))
    assert np.isclose(m[1], 0.09 * gt.mean())


def test_valid_depth_filter_crop_scale_clip():
    h, w = 100, 200
    rng = np.random.default_rng(0)
    true = rng.uniform(2, 60, (h, w))
    true[:, :5] = 0                                         # invalid lidar pixels
    true[50, 100] = opts.MAX_DEPTH + 10
    pred = true * 0.5 + 1e-3
    pred[45, 50] = 1e6
    p, t = eu.valid_depth_filter(pred[..., None], true[..., None])
    r0, r1, c0, c1 = int(0.40810811 * h), int(0.99189189 * h), int(0.03594771 * w), int(0.96405229 * w)
    crop = np.zeros((h, w), bool)
    crop[r0:r1, c0:c1] = True
    mask = crop & (true > opts.MIN_DEPTH) & (true < opts.MAX_DEPTH)
    assert p.shape == t.shape == (mask.sum(),)
    assert np.array_equal(t, true[mask])
    assert np.isclose(np.median(p), np.median(t), rtol=1e-3)     # median scaling
    assert p.max() <= opts.MAX_DEPTH and p.min() >= opts.MIN_DEPTH
    assert pred[45, 50] == 1e6                                    # input untouched
    m = eu.evaluate_depth([pred, pred], [true, true])
    assert m.shape == (7,) and m[0] < 0.01


def test_pose_metric_trajectory_error():
    """eval_utils.py:157-192: doubling the translation offset doubles the absolute error; a pure scale change of the
    translations has zero scale-aligned error."""
    rng = np.random.default_rng(1)
    v1 = rng.random((8, 4, 6)) * 2.0 - 1.0
    v2, v3, v4 = v1.copy(), v1.copy(), v1.copy()
    v2[:, 1:] += np.array([0, 1, 0, 0, 0, 0])
    v3[:, 1:] += np.array([0, 2, 0, 0, 0, 0])
    v4[:, :, :3] *= 2.0
    e12, e13, e14 = eu.PoseMetricNumpy(), eu.PoseMetricNumpy(), eu.PoseMetricNumpy()
    e12.compute_pose_errors(v1, eu.pose_rvec2matr_batch_np(v2))
    e13.compute_pose_errors(v1, eu.pose_rvec2matr_batch_np(v3))
    e14.compute_pose_errors(v1, eu.pose_rvec2matr_batch_np(v4))
    assert e12.trj_abs_err.shape == (8, 4)
    assert np.isclose(e12.trj_abs_err * 2.0, e13.trj_abs_err, atol=1e-5).all()
    assert np.isclose(e14.trj_rel_err, 0, atol=1e-5).all()


def test_pose_metric_rotational_error():
    """eval_utils.py:195-221: unit rotations about one axis scaled by (1, 0.5, 1, 1.5) -> errors (0.5, 0, 0, 0.5)."""
    rng = np.random.default_rng(2)
    v1 = rng.random((8, 4, 6)) * 2.0 - 1.0
    v1[:, 1:, 3:] = v1[:, 0:1, 3:]
    v1[:, :, 3:] /= np.linalg.norm(v1[:, 0:1, 3:], axis=2, keepdims=True)
    v2 = v1.copy()
    v2[:, 1, 3:] *= 0.5
    v2[:, 3, 3:] *= 1.5
    e = eu.PoseMetricNumpy()
    e.compute_pose_errors(v1, eu.pose_rvec2matr_batch_np(v2))
    assert np.isclose(e.rot_err[:, 0], 0.5).all()
    assert np.isclose(e.rot_err[:, 1], 0.0, atol=1e-3).all()
    assert np.isclose(e.rot_err[:, 2], 0.0, atol=1e-3).all()
    assert np.isclose(e.rot_err[:, 3], 0.5).all()
    assert len(e.get_mean_pose_error()) == 3


def test_pose_rvec2matr_np_matches_oracle():
    import torch
    from oracle import ref_pose
    v = np.random.default_rng(3).random((3, 4, 6)) * 2 - 1
    a = eu.pose_rvec2matr_batch_np(v)
    b = ref_pose.pose_rvec2matr_batch(torch.from_numpy(v)).numpy()
    assert np.allclose(a, b, atol=1e-10)
    assert np.allclose(eu.pose_rvec2matr_batch_np(np.zeros((1, 1, 6)))[0, 0], np.eye(4))
