"""oracle (test infrastructure): differentiable view synthesis.

Follows model/synthesize/synthesize_base.py:10-178 (SynthesizeMultiScale /
SynthesizeSingleScale) and model/synthesize/bilinear_interp.py:5-147
(BilinearInterpolation) of the reference, op for op, on PyTorch-CPU tensors so
that torch.autograd reproduces the TF gradient (floor / clip / equal carry no
gradient; the gradient reaches the coordinates through the weights only).
Axis order is the reference's: images [batch, numsrc, height, width, C].
"""
import torch
import torch.nn.functional as F

from .ref_pose import pose_rvec2matr_batch


# --------------------------------------------------------------------------- helpers
def tf_resize_bilinear(img_nhwc, size):
    """tf.image.resize(..., method="bilinear") of TF2: half-pixel centres, no antialias.
    img_nhwc: [M, H, W, C] -> [M, size[0], size[1], C]."""
    x = img_nhwc.permute(0, 3, 1, 2)
    x = F.interpolate(x, size=tuple(size), mode="bilinear", align_corners=False, antialias=False)
    return x.permute(0, 2, 3, 1)


# --------------------------------------------------------------------------- bilinear_interp.py
def neighbor_int_pixels(pixel_coords, height, width):
    """bilinear_interp.py:34-50 -> (u_floor, u_ceil, v_floor, v_ceil) [B, N, 4, H*W]."""
    u = pixel_coords[:, :, 0:1, :]
    u_floor = torch.floor(u)
    u_ceil = torch.clamp(u_floor + 1, 0, width - 1)
    u_floor = torch.clamp(u_floor, 0, width - 1)
    v = pixel_coords[:, :, 1:2, :]
    v_floor = torch.floor(v)
    v_ceil = torch.clamp(v_floor + 1, 0, height - 1)
    v_floor = torch.clamp(v_floor, 0, height - 1)
    return torch.cat([u_floor, u_ceil, v_floor, v_ceil], dim=2)


def make_valid_mask(pixel_floorceil, valid_mask, batch):
    """bilinear_interp.py:53-77 -> [B, N, 1, H*W] float mask."""
    uf = pixel_floorceil[:, :, 0:1, :]
    uc = pixel_floorceil[:, :, 1:2, :]
    vf = pixel_floorceil[:, :, 2:3, :]
    vc = pixel_floorceil[:, :, 3:4, :]
    mask = torch.logical_and(uf + 1 == uc, vf + 1 == vc)
    if valid_mask is not None:
        nonzero_mask = valid_mask.reshape(batch, 1, 1, -1) != 0
        mask = torch.logical_and(mask, nonzero_mask)
    return mask.to(pixel_floorceil.dtype)


def calc_neighbor_weights(pixel_coords, pixel_floorceil, valid_mask):
    """bilinear_interp.py:80-102 -> (w_uf_vf, w_uf_vc, w_uc_vf, w_uc_vc) [B, N, 4, H*W]."""
    w_uf = pixel_floorceil[:, :, 1:2, :] - pixel_coords[:, :, 0:1, :]
    w_uc = pixel_coords[:, :, 0:1, :] - pixel_floorceil[:, :, 0:1, :]
    w_vf = pixel_floorceil[:, :, 3:4, :] - pixel_coords[:, :, 1:2, :]
    w_vc = pixel_coords[:, :, 1:2, :] - pixel_floorceil[:, :, 2:3, :]
    weights = torch.cat([w_uf * w_vf, w_uf * w_vc, w_uc * w_vf, w_uc * w_vc], dim=2)
    return weights * valid_mask


def sample_neighbor_images(source_image, pixel_floorceil):
    """bilinear_interp.py:105-132: four gather_nd(batch_dims=2) -> [B, N, 4, H*W, C]."""
    batch, numsrc, height, width, channels = source_image.shape
    idx = pixel_floorceil.detach().to(torch.int64)
    uf, uc, vf, vc = idx[:, :, 0], idx[:, :, 1], idx[:, :, 2], idx[:, :, 3]
    flat = source_image.reshape(batch, numsrc, height * width, channels)

    def gather(v, u):
        lin = (v * width + u).unsqueeze(-1).expand(-1, -1, -1, channels)
        return torch.gather(flat, 2, lin)

    return torch.stack([gather(vf, uf), gather(vc, uf), gather(vf, uc), gather(vc, uc)], dim=2)


def merge_images(sampled_images, weights):
    """bilinear_interp.py:134-146."""
    return torch.sum(sampled_images * weights.unsqueeze(-1), dim=2)


def bilinear_interpolation(image, pixel_coords, valid_mask=None):
    """BilinearInterpolation.__call__, bilinear_interp.py:7-32.

    image [B, N, H, W, C]; pixel_coords (u, v[, 1]) [B, N, 2 or 3, H*W];
    valid_mask [B, H, W, 1] or None -> [B, N, H, W, C]."""
    batch, numsrc, height, width, channels = image.shape
    pixel_floorceil = neighbor_int_pixels(pixel_coords, height, width)
    mask = make_valid_mask(pixel_floorceil, valid_mask, batch)
    weights = calc_neighbor_weights(pixel_coords, pixel_floorceil, mask)
    sampled = sample_neighbor_images(image, pixel_floorceil)
    flat_image = merge_images(sampled, weights)
    return flat_image.reshape(batch, numsrc, height, width, channels)


def flow_to_pixel_coordinates(flow):
    """FlowBilinearInterpolation.flow_to_pixel_coordinates, bilinear_interp.py:183-205.
    flow [B, N, H, W, 2(u,v)] -> coords [B, N, 2, H*W] = grid - flow."""
    batch, numsrc, height, width, _ = flow.shape
    u = torch.arange(0, width, dtype=flow.dtype)
    v = torch.arange(0, height, dtype=flow.dtype)
    vgrid, ugrid = torch.meshgrid(v, u, indexing="ij")
    uvgrid = torch.stack([ugrid, vgrid], dim=0).reshape(1, 1, 2, -1)
    uvflow = flow.reshape(batch, numsrc, -1, 2).permute(0, 1, 3, 2)
    return uvgrid - uvflow


# --------------------------------------------------------------------------- synthesize_base.py
def scale_intrinsic(intrinsic, scale):
    """synthesize_base.py:66-71: rows 0,1 divided by scale, row 2 = (0,0,1)."""
    batch = intrinsic.shape[0]
    scaled_part = intrinsic[:, :2, :] / scale
    const_part = torch.tensor([[[0, 0, 1]]], dtype=intrinsic.dtype).expand(batch, 1, 3)
    return torch.cat([scaled_part, const_part], dim=1)


def resize_source_images(source_image, height_sc, width_sc):
    """synthesize_base.py:74-85."""
    batch, numsrc, height, width, ch = source_image.shape
    x = source_image.reshape(batch * numsrc, height, width, ch)
    x = tf_resize_bilinear(x, (height_sc, width_sc))
    return x.reshape(batch, numsrc, height_sc, width_sc, ch)


def pixel_meshgrid(height, width, dtype=torch.float32):
    """synthesize_base.py:114-124 -> (u, v, 1) [3, H*W], row-major pixel order."""
    v = torch.linspace(0, height - 1, height, dtype=dtype)
    u = torch.linspace(0, width - 1, width, dtype=dtype)
    vgrid, ugrid = torch.meshgrid(v, u, indexing="ij")
    uv = torch.stack([ugrid, vgrid], dim=0).reshape(2, -1)
    return torch.cat([uv, torch.ones((1, height * width), dtype=dtype)], dim=0)


def pixel2cam(pixel_coords, depth, intrinsic):
    """synthesize_base.py:126-146 -> homogeneous target-frame points [B, 4, H*W]."""
    batch = intrinsic.shape[0]
    depth = depth.reshape(batch, 1, -1)
    cam_coords = torch.tensordot(torch.linalg.inv(intrinsic), pixel_coords, dims=([2], [0]))
    cam_coords = cam_coords * depth
    num_pts = cam_coords.shape[2]
    return torch.cat([cam_coords, torch.ones((batch, 1, num_pts), dtype=cam_coords.dtype)], dim=1)


def transform_to_source(tgt_coords, t2s_pose):
    """synthesize_base.py:149-159 -> [B, N, 4, H*W]."""
    numsrc = t2s_pose.shape[1]
    tgt = tgt_coords.unsqueeze(1).expand(-1, numsrc, -1, -1)
    return torch.matmul(t2s_pose, tgt)


def cam2pixel(cam_coords, intrinsic):
    """synthesize_base.py:161-178 -> (u, v, ~1) [B, N, 3, H*W]; p / (p_z + 1e-10)."""
    numsrc = cam_coords.shape[1]
    intrinsic_expand = intrinsic.unsqueeze(1).expand(-1, numsrc, -1, -1)
    point_coords = cam_coords[:, :, :3, :]
    pixel_coords = torch.matmul(intrinsic_expand, point_coords)
    pixel_scales = pixel_coords[:, :, 2:3, :]
    return pixel_coords / (pixel_scales + 1e-10)


def warp_pixel_coords(tgt_depth, pose, intrinsic, height, width):
    """synthesize_base.py:106-112."""
    tgt_pixel_coords = pixel_meshgrid(height, width, dtype=tgt_depth.dtype)
    tgt_cam_coords = pixel2cam(tgt_pixel_coords, tgt_depth, intrinsic)
    src_cam_coords = transform_to_source(tgt_cam_coords, pose)
    return cam2pixel(src_cam_coords, intrinsic)


def synthesize_single_scale(source_image, intrinsic, depth_sc, poses_matr):
    """SynthesizeSingleScale.__call__, synthesize_base.py:39-58."""
    height_orig = source_image.shape[2]
    _, height_sc, width_sc, _ = depth_sc.shape
    scale = int(height_orig // height_sc)                              # (:64)
    intrinsic_sc = scale_intrinsic(intrinsic, scale)
    source_images_sc = resize_source_images(source_image, height_sc, width_sc)
    coords = warp_pixel_coords(depth_sc, poses_matr, intrinsic_sc, height_sc, width_sc)
    return bilinear_interpolation(source_images_sc, coords, depth_sc)


def synthesize_multi_scale(source_image, intrinsic, pred_depth_ms, pred_pose):
    """SynthesizeMultiScale.__call__, synthesize_base.py:10-29.

    source_image [B, N, H, W, 3]; intrinsic [B, 3, 3]; pred_depth_ms list of
    [B, H/s, W/s, 1]; pred_pose twist [B, N, 6] (target -> source)
    -> list of [B, N, H/s, W/s, 3]."""
    poses_matr = pose_rvec2matr_batch(pred_pose)
    return [synthesize_single_scale(source_image, intrinsic, depth_sc, poses_matr)
            for depth_sc in pred_depth_ms]
