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
