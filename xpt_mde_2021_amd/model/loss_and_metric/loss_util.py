"""photometric_loss_l1 / _l2 / _ssim with the reference's signatures (model/loss_and_metric/loss_util.py:6-96),
backed by the gfx950 photometric kernels (K4, K5; forward and backward)."""
from ...hip import ops as _ops


def photometric_loss_l1(synt_target, orig_target, reduce=True):
    """synt_target [batch, numsrc, h, w, 3], orig_target [batch, h, w, 3] -> [batch] (or the per-pixel
    [batch, numsrc, h, w, 3] map when reduce=False).  Black synthesized pixels (mean_c == 0) contribute 0 but
    stay in the mean's denominator (loss_util.py:15-24)."""
    return _ops.photometric("L1", synt_target, orig_target, reduce)


def photometric_loss_l2(synt_target, orig_target, reduce=True):
    """loss_util.py:29-48."""
    return _ops.photometric("L2", synt_target, orig_target, reduce)


def photometric_loss_ssim(synt_target, orig_target, reduce=True):
    """loss_util.py:52-96: clip((1 - SSIM_3x3)/2, 0, 1) with SAME average pooling whose divisor excludes
    the padding, c1 = 0.01^2, c2 = 0.03^2."""
    return _ops.photometric("SSIM", synt_target, orig_target, reduce)
