"""oracle (test infrastructure): photometric / SSIM / smoothness losses and TotalLoss.

Follows model/loss_and_metric/loss_util.py:6-96, model/loss_and_metric/losses.py:14-154,
175-232, 282-321, 377-494 and utils/util_funcs.py:146-175 of the reference on
PyTorch-CPU tensors (same op decomposition as the TF graph).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import ref_pose as cp
from .ref_synthesize import synthesize_multi_scale, tf_resize_bilinear

IMAGE_GRADIENT_FACTOR = 4          # config-example.py:67


# --------------------------------------------------------------------------- utils/util_funcs.py
def safe_reciprocal_number(src):
    """util_funcs.py:157-160: (1/x) * [x > 1e-5]."""
    mask = (src > 0.00001).to(src.dtype)
    return (1. / src) * mask


def safe_reciprocal_number_ms(src_ms):
    """util_funcs.py:146-154."""
    return [safe_reciprocal_number(src) for src in src_ms]


def multi_scale_like_depth(image, depth_ms):
    """util_funcs.py:163-175: TF2 bilinear resize of image [B,H,W,3] to every depth scale."""
    return [tf_resize_bilinear(image, depth.shape[1:3]) for depth in depth_ms]


# --------------------------------------------------------------------------- loss_util.py
def photometric_loss_l1(synt_target, orig_target, reduce=True):
    """loss_util.py:6-25.  synt [B,N,h,w,3], orig [B,h,w,3] -> [B] (or per-pixel map)."""
    orig_target = orig_target.unsqueeze(1)
    synt_target_gray = torch.mean(synt_target, dim=-1, keepdim=True)
    error_mask = synt_target_gray == 0
    photo_error = torch.abs(synt_target - orig_target)
    photo_error = torch.where(error_mask, torch.zeros((), dtype=photo_error.dtype), photo_error)
    if reduce:
        photo_error = torch.mean(photo_error, dim=[1, 2, 3, 4])
    return photo_error


def photometric_loss_l2(synt_target, orig_target, reduce=True):
    """loss_util.py:29-48."""
    orig_target = orig_target.unsqueeze(1)
    synt_target_gray = torch.mean(synt_target, dim=-1, keepdim=True)
    error_mask = synt_target_gray == 0
    photo_error = torch.square(synt_target - orig_target)
    photo_error = torch.where(error_mask, torch.zeros((), dtype=photo_error.dtype), photo_error)
    if reduce:
        photo_error = torch.mean(photo_error, dim=[1, 2, 3, 4])
    return photo_error


def average_pool_3x3_same(x):
    """tf.keras.layers.AveragePooling3D(pool=(1,3,3), strides=1, padding="SAME") on
    [B, N, h, w, C] (loss_util.py:78): 3x3 spatial mean whose divisor EXCLUDES the padding."""
    b, n, h, w, c = x.shape
    y = x.permute(0, 1, 4, 2, 3).reshape(b * n * c, 1, h, w)
    y = F.avg_pool2d(y, kernel_size=3, stride=1, padding=1, count_include_pad=False)
    return y.reshape(b, n, c, h, w).permute(0, 1, 3, 4, 2)


def photometric_loss_ssim(synt_target, orig_target, reduce=True):
    """loss_util.py:52-96."""
    numsrc = synt_target.shape[1]
    orig_target = orig_target.unsqueeze(1).repeat(1, numsrc, 1, 1, 1)     # tf.tile (:61)
    synt_target_gray = torch.mean(synt_target, dim=-1, keepdim=True)
    error_mask = synt_target_gray == 0

    x = orig_target
    y = synt_target
    c1 = 0.01 ** 2
    c2 = 0.03 ** 2
    mu_x = average_pool_3x3_same(x)
    mu_y = average_pool_3x3_same(y)
    sigma_x = average_pool_3x3_same(x ** 2) - mu_x ** 2
    sigma_y = average_pool_3x3_same(y ** 2) - mu_y ** 2
    sigma_xy = average_pool_3x3_same(x * y) - mu_x * mu_y

    ssim_n = (2 * mu_x * mu_y + c1) * (2 * sigma_xy + c2)
    ssim_d = (mu_x ** 2 + mu_y ** 2 + c1) * (sigma_x + sigma_y + c2)
    ssim = ssim_n / ssim_d
    ssim = torch.clamp((1 - ssim) / 2, 0, 1)
    ssim = torch.where(error_mask, torch.zeros((), dtype=ssim.dtype), ssim)
    if reduce:
        ssim = torch.mean(ssim, dim=[1, 2, 3, 4])
    return ssim


PHOTOMETRIC = {"L1": photometric_loss_l1, "L2": photometric_loss_l2, "SSIM": photometric_loss_ssim}


# --------------------------------------------------------------------------- losses.py: per-type losses
def merge_multi_scale_losses(losses, scale_weights):
    """LossBase.merge_multi_scale_losses, losses.py:147-154: [S,B]^T @ w[S,1] -> [B,1]."""
    stacked = torch.stack(losses, dim=0)
    sw = torch.as_tensor(np.asarray(scale_weights, dtype=np.float64).reshape(-1, 1), dtype=stacked.dtype)
    return torch.matmul(stacked.t(), sw)


def photometric_loss_multi_scale(method, synth_target_ms, target_ms, scale_weights):
    """PhotometricLossMultiScale.__call__, losses.py:179-195."""
    fn = PHOTOMETRIC[method]
    losses = [fn(s, t) for s, t in zip(synth_target_ms, target_ms)]
    return merge_multi_scale_losses(losses, scale_weights)


def smootheness_loss(disp, image, grad_factor=IMAGE_GRADIENT_FACTOR):
    """SmoothenessLossMultiScale.smootheness_loss, losses.py:409-440.  disp [B,h,w,1], image [B,h,w,3] -> [B]."""
    def gradient_x(img):
        return img[:, :, :-1, :] - img[:, :, 1:, :]

    def gradient_y(img):
        return img[:, :-1, :, :] - img[:, 1:, :, :]

    disp_gradients_x = gradient_x(disp)
    disp_gradients_y = gradient_y(disp)
    image_gradients_x = gradient_x(image)
    image_gradients_y = gradient_y(image)
    weights_x = torch.exp(-torch.mean(torch.abs(image_gradients_x * grad_factor), 3, keepdim=True))
    weights_y = torch.exp(-torch.mean(torch.abs(image_gradients_y * grad_factor), 3, keepdim=True))
    smoothness_x = disp_gradients_x * weights_x
    smoothness_y = disp_gradients_y * weights_y
    smoothness_x = 0.5 * torch.mean(torch.abs(smoothness_x), dim=[1, 2, 3])
    smoothness_y = 0.5 * torch.mean(torch.abs(smoothness_y), dim=[1, 2, 3])
    return smoothness_x + smoothness_y


def smootheness_loss_multi_scale(disp_ms, target_ms, scale_weights):
    """SmoothenessLossMultiScale.__call__, losses.py:391-407 (each scale divided by its scale)."""
    losses = []
    orig_width = target_ms[0].shape[2]
    for disp, image in zip(disp_ms, target_ms):
        scale = orig_width / image.shape[2]
        losses.append(smootheness_loss(disp, image) / scale)
    return merge_multi_scale_losses(losses, scale_weights)


def resize_bilinear_5d(srcimg, dst_hw):
    """losses.py:377-383."""
    b, n, hs, ws, c = srcimg.shape
    dst = tf_resize_bilinear(srcimg.reshape(b * n, hs, ws, c), dst_hw)
    return dst.reshape(b, n, dst_hw[0], dst_hw[1], c)


def monodepth2_loss_multi_scale(method, synth_target_ms, original_target, scale_weights):
    """MonoDepth2LossMultiScale.__call__, losses.py:205-232: upsample synth, min over sources."""
    fn = PHOTOMETRIC[method]
    ho, wo = original_target.shape[1:3]
    losses = []
    for synt in synth_target_ms:
        synt_rsz = resize_bilinear_5d(synt, (ho, wo))
        loss = fn(synt_rsz, original_target, False)
        loss = torch.min(loss, dim=1).values
        losses.append(torch.mean(loss, dim=[1, 2, 3]))
    return merge_multi_scale_losses(losses, scale_weights)


def moa_loss_multi_scale(method, temp_synth_ms, stereo_synth_ms, original_target, scale_weights):
    """MoALossMultiScale.__call__, losses.py:289-321: min over temporal + stereo views."""
    fn = PHOTOMETRIC[method]
    ho, wo = original_target.shape[1:3]
    losses = []
    for temp_target, stro_target in zip(temp_synth_ms, stereo_synth_ms):
        temp_loss = fn(resize_bilinear_5d(temp_target, (ho, wo)), original_target, False)
        stro_loss = fn(resize_bilinear_5d(stro_target, (ho, wo)), original_target, False)
        moa = torch.cat([temp_loss, stro_loss], dim=1)
        moa = torch.min(moa, dim=1).values
        losses.append(torch.mean(moa, dim=[1, 2, 3]))
    return merge_multi_scale_losses(losses, scale_weights)


def md2comb_loss_multi_scale(method, synth_target_ms, warped_target_ms, original_target, scale_weights):
    """MD2CombLossMultiScale.__call__, losses.py:324-374: static loss + 1000 where it exceeds twice the flow loss (finest
    flow scale, both at the original size), min over sources, sum over the kept (< 1000) elements of each sample divided by
    the kept-element count of the whole [B,H,W,3] tensor (tf.math.count_nonzero without axis, :369)."""
    fn = PHOTOMETRIC[method]
    ho, wo = original_target.shape[1:3]
    flow_loss = fn(resize_bilinear_5d(warped_target_ms[0], (ho, wo)), original_target, False)
    losses = []
    for synt in synth_target_ms:
        static_loss = fn(resize_bilinear_5d(synt, (ho, wo)), original_target, False)
        mask = (static_loss > flow_loss * 2.0).to(static_loss.dtype)
        static_loss = static_loss + mask * 1000.0
        static_loss = torch.min(static_loss, dim=1).values
        keep = (static_loss < 1000.0).to(static_loss.dtype)
        losses.append(torch.sum(static_loss * keep, dim=[1, 2, 3]) / torch.count_nonzero(keep).to(static_loss.dtype))
    return merge_multi_scale_losses(losses, scale_weights)


def stereo_depth_loss(method, augm_data, scale_weights):
    """StereoDepthLoss.__call__, losses.py:447-478: left + right per scale, then scale merge."""
    fn = PHOTOMETRIC[method]
    left = [fn(s, t) for s, t in zip(augm_data["stereo_synth_ms"], augm_data["target_ms"])]
    right = [fn(s, t) for s, t in zip(augm_data["stereo_synth_ms_R"], augm_data["target_ms_R"])]
    losses = [l + r for l, r in zip(left, right)]
    return merge_multi_scale_losses(losses, scale_weights)


def stereo_pose_loss(features, predictions):
    """StereoPoseLoss.__call__, losses.py:481-494."""
    pose_lr_true_mat = features["stereo_T_LR"].unsqueeze(1)
    pose_rl_true_mat = torch.linalg.inv(pose_lr_true_mat)
    pose_lr_true = cp.pose_matr2rvec_batch(pose_lr_true_mat)
    pose_rl_true = cp.pose_matr2rvec_batch(pose_rl_true_mat)
    mse_lr = torch.mean(torch.square(pose_lr_true - predictions["pose_LR"]), dim=-1)
    mse_rl = torch.mean(torch.square(pose_rl_true - predictions["pose_RL"]), dim=-1)
    return torch.mean(mse_lr + mse_rl, dim=1)


# --------------------------------------------------------------------------- losses.py: TotalLoss
def append_data(features, predictions, suffix=""):
    """TotalLoss.append_data, losses.py:57-104 (depth/pose branch; flow branch is out of scope)."""
    image5d = features["image5d" + suffix]
    intrinsic = features["intrinsic" + suffix]
    source_image = image5d[:, :-1]
    target_image = image5d[:, -1]
    augm = {"source" + suffix: source_image, "target" + suffix: target_image}
    if ("depth_ms" + suffix in predictions) and ("pose" + suffix in predictions):
        pred_depth_ms = predictions["depth_ms" + suffix]
        pred_pose = predictions["pose" + suffix]
        augm["target_ms" + suffix] = multi_scale_like_depth(target_image, pred_depth_ms)
        augm["synth_target_ms" + suffix] = synthesize_multi_scale(source_image, intrinsic, pred_depth_ms, pred_pose)
    return augm


def synethesize_stereo(features, predictions, augm_data):
    """TotalLoss.synethesize_stereo (sic), losses.py:106-140."""
    out = {}
    if ("stereo_T_LR" not in features) or ("depth_ms" not in predictions):
        return out
    pose_T_RL = torch.linalg.inv(features["stereo_T_LR"])
    pose_T_RL = cp.pose_matr2rvec_batch(pose_T_RL.unsqueeze(1))
    out["stereo_synth_ms"] = synthesize_multi_scale(augm_data["target_R"].unsqueeze(1), features["intrinsic"],
                                                    predictions["depth_ms"], pose_T_RL)
    pose_T_LR = cp.pose_matr2rvec_batch(features["stereo_T_LR"].unsqueeze(1))
    out["stereo_synth_ms_R"] = synthesize_multi_scale(augm_data["target"].unsqueeze(1), features["intrinsic"],
                                                      predictions["depth_ms_R"], pose_T_LR)
    return out


def loss_by_name(name, features, predictions, augm, scale_weights):
    """loss_factory.py:9-37 pool, restricted to the in-scope rigid losses."""
    sfx = "_R" if name.endswith("_R") else ""
    base = name[:-2] if sfx else name
    if base in ("L1", "SSIM"):
        return photometric_loss_multi_scale(base, augm["synth_target_ms" + sfx], augm["target_ms" + sfx], scale_weights)
    if base == "smoothe":
        return smootheness_loss_multi_scale(predictions["disp_ms" + sfx], augm["target_ms" + sfx], scale_weights)
    if base in ("md2L1", "md2SSIM"):
        return monodepth2_loss_multi_scale(base[3:], augm["synth_target_ms" + sfx], augm["target" + sfx], scale_weights)
    if base in ("moaL1", "moaSSIM"):
        return moa_loss_multi_scale(base[3:], augm["synth_target_ms" + sfx], augm["stereo_synth_ms"],
                                    augm["target" + sfx], scale_weights)
    if base in ("stereoL1", "stereoSSIM"):
        return stereo_depth_loss(base[6:], augm, scale_weights)
    if base == "stereoPose":
        return stereo_pose_loss(features, predictions)
    raise KeyError(name)


def total_loss(predictions, features, loss_weights, scale_weights, stereo, batch_size):
    """TotalLoss.__call__, losses.py:26-55 -> (total scalar, {name: unweighted mean}).

    loss_weights must already be filtered as loss_factory.py:41-47 does (zero weights and
    losses whose dataset keys are missing are dropped)."""
    augm = append_data(features, predictions)
    if stereo and ("image_R" in features or "image5d_R" in features):
        augm.update(append_data(features, predictions, "_R"))
        augm.update(synethesize_stereo(features, predictions, augm))
    total = 0.
    by_type = {}
    for name, weight in loss_weights.items():
        loss_batch = loss_by_name(name, features, predictions, augm, scale_weights)
        loss_mean = torch.sum(loss_batch) / batch_size          # tf.nn.compute_average_loss (:49)
        total = total + loss_mean * weight
        by_type[name] = loss_mean
    return total, by_type
