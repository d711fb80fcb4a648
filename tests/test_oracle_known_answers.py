"""Pins the oracle (CPU restatement) with the reference's own data-free known-answer tests.

Each test restates the inputs and the asserted facts of one reference test:
  model/synthesize/test_synthesizing.py:149-301, utils/convert_pose.py:222-271,
  utils/tests.py:62-76, model/loss_and_metric/losses.py:541-559.
"""
import numpy as np
import torch

from oracle import ref_loss, ref_pose, ref_synthesize as rs


def test_scale_intrinsic():
    # test_synthesizing.py:149-163
    batch = 8
    intrinsic = torch.tensor([8, 0, 4, 0, 8, 4, 0, 0, 1], dtype=torch.float32).reshape(1, 3, 3).repeat(batch, 1, 1)
    out = rs.scale_intrinsic(intrinsic, 2)
    assert np.allclose(intrinsic[:, :2, :] / 2, out[:, :2, :])
    assert np.allclose(intrinsic[:, -1, :], out[:, -1, :])


def test_pixel2cam():
    # test_synthesizing.py:166-183 (+ the by-hand values SURVEY 8c lists)
    batch, height, width = 8, 4, 4
    pix = rs.pixel_meshgrid(height, width)
    intrinsic = torch.tensor([4, 0, height / 2, 0, 4, width / 2, 0, 0, 1], dtype=torch.float32)
    intrinsic = intrinsic.reshape(1, 3, 3).repeat(batch, 1, 1)
    depth = torch.ones((batch, height, width)) * 2
    cam = rs.pixel2cam(pix, depth, intrinsic)
    assert tuple(cam.shape) == (batch, 4, height * width)
    u = pix[0].numpy()
    v = pix[1].numpy()
    assert np.allclose(cam[0, 0], (u - 2) / 4 * 2)
    assert np.allclose(cam[0, 1], (v - 2) / 4 * 2)
    assert np.allclose(cam[0, 2], 2)
    assert np.allclose(cam[0, 3], 1)


def test_transform_to_source():
    # test_synthesizing.py:186-208
    batch, num_pts, numsrc = 8, 6, 3
    coords = np.arange(1, 4 * num_pts + 1).reshape((num_pts, 4)).T.copy()
    coords[3, :] = 1
    coords = np.tile(coords, (batch, 1, 1))
    poses = np.identity(4) * 2
    poses[:3, 3] = 1
    poses[3, 3] = 1
    poses = np.tile(poses, (batch, numsrc, 1, 1))
    src = rs.transform_to_source(torch.tensor(coords, dtype=torch.float32), torch.tensor(poses, dtype=torch.float32))
    assert np.allclose(coords[2, :3] * 2 + 1, src[2, 1, :3])


def test_pixel_weighting():
    # test_synthesizing.py:211-255
    rng = np.random.default_rng(0)
    batch, numsrc, height, width = 8, 4, 5, 5
    pc = rng.uniform(0.1, 3.9, (batch, numsrc, 3, height * width))
    pc[:, :, :, 0] = -1.5
    pc[:, :, :, 1] = 7
    chk_u, chk_v = 0.2, 0.7
    pc[:, :, 0, 3] = 2 + chk_u
    pc[:, :, 1, 3] = 3 + chk_v
    pc[:, :, 2, :] = 1
    pc_t = torch.tensor(pc, dtype=torch.float32)
    fc = rs.neighbor_int_pixels(pc_t, height, width)
    assert np.allclose(np.floor(pc[:, :, 0, 2:]), fc[:, :, 0, 2:])
    assert np.allclose(np.ceil(pc[:, :, 1, 2:]), fc[:, :, 3, 2:])
    mask = rs.make_valid_mask(fc, None, batch)
    weights = rs.calc_neighbor_weights(pc_t, fc, mask)
    assert np.allclose(weights[:, :, 0, 3], (1 - chk_u) * (1 - chk_v), atol=1e-6)
    assert np.allclose(weights[:, :, 1, 3], (1 - chk_u) * chk_v, atol=1e-6)
    assert np.allclose(weights[:, :, 2, 3], chk_u * (1 - chk_v), atol=1e-6)
    assert np.allclose(weights[:, :, 3, 3], chk_u * chk_v, atol=1e-6)
    wsum = weights.sum(dim=2).numpy()
    assert (np.isclose(wsum, 0, atol=1e-6) | np.isclose(wsum, 1, atol=1e-6)).all()
    # columns 0 (-1.5) and 1 (7) are out of the image -> all-zero weights
    assert np.all(weights[:, :, :, 0].numpy() == 0)
    assert np.all(weights[:, :, :, 1].numpy() == 0)


def test_reconstruct_bilinear_interp():
    # test_synthesizing.py:258-301
    batch, numsrc, height, width = 8, 4, 5, 5
    pc = np.meshgrid(np.arange(0, height), np.arange(0, width))
    pc = np.stack(pc, axis=0).reshape((1, 1, 2, 5, 5)).astype(np.float32)
    u_add = 1.3
    pc[0, 0, 0] += u_add
    pc = np.tile(pc, (batch, numsrc, 1, 1, 1)).reshape((batch, numsrc, 2, height * width))
    pc_t = torch.tensor(pc)
    fc = rs.neighbor_int_pixels(pc_t, height, width)
    mask = rs.make_valid_mask(fc, None, batch)
    expected_mask = np.zeros((batch, numsrc, height, width), dtype=np.float64)
    expected_mask[:, :, :4, :3] = 1
    assert np.allclose(expected_mask.reshape((batch, numsrc, 1, height * width)), mask)

    image = np.meshgrid(np.arange(0, height), np.arange(0, width))[0].reshape((1, 1, height, width, 1))
    image = np.tile(image, (batch, numsrc, 1, 1, 3)).astype(np.float32)
    depth = np.ones((batch, height, width, 1), dtype=np.float32)
    recon = rs.bilinear_interpolation(torch.tensor(image), pc_t, torch.tensor(depth))
    expected = (image + u_add) * expected_mask.reshape((batch, numsrc, height, width, 1))
    assert np.allclose(recon, expected, atol=1e-5)


def test_pose_rvec2matr_batch():
    # convert_pose.py:222-242
    g = torch.Generator().manual_seed(1)
    poses = torch.rand((8, 4, 6), generator=g) * 2 - 1
    matr = ref_pose.pose_rvec2matr_batch(poses)
    pose0 = poses[3, 2].numpy()
    matr0 = matr[3, 2].numpy()
    assert np.allclose(pose0[:3], matr0[:3, 3])
    angle_mat = np.arccos((np.trace(matr0[:3, :3]) - 1) / 2)
    assert np.isclose(np.linalg.norm(pose0[3:]), angle_mat, atol=1e-5)
    assert np.allclose(matr[:, :, 3], np.array([0, 0, 0, 1.]))


def test_pose_matr2rvec_batch_round_trip():
    # convert_pose.py:256-271
    g = torch.Generator().manual_seed(2)
    twist = torch.rand((8, 4, 6), generator=g) * 2 - 1
    again = ref_pose.pose_matr2rvec_batch(ref_pose.pose_rvec2matr_batch(twist))
    assert np.allclose(twist.numpy(), again.numpy(), atol=1e-5, rtol=1e-4)


def test_rotation_convention_negated_skew():
    # utils/tests.py:62-76: rvec = (0, 0, pi/3) -> R = [[c, s, 0], [-s, c, 0], [0, 0, 1]]
    angle = np.pi / 3
    pose = torch.tensor([[[0, 0, 0, 0, 0, angle]]], dtype=torch.float64)
    R = ref_pose.pose_rvec2matr_batch(pose)[0, 0, :3, :3].numpy()
    c, s = np.cos(angle), np.sin(angle)
    assert np.allclose(R, np.array([[c, s, 0], [-s, c, 0], [0, 0, 1]]))


def test_average_pool_3d_interior_and_border():
    # losses.py:541-559: pooled[0,0,11,11,1] == mean(x[0,0,10:13,10:13,1]);
    # SAME padding excludes the padding from the divisor (TF semantics) -> corners average 4 values.
    g = torch.Generator().manual_seed(3)
    x = torch.randn((2, 4, 30, 30, 3), generator=g)
    mu = ref_loss.average_pool_3x3_same(x)
    assert np.isclose(x[0, 0, 10:13, 10:13, 1].mean().item(), mu[0, 0, 11, 11, 1].item(), atol=1e-6)
    assert np.isclose(x[1, 2, 0:2, 0:2, 0].mean().item(), mu[1, 2, 0, 0, 0].item(), atol=1e-6)
    assert np.isclose(x[1, 2, 0:2, 4:7, 2].mean().item(), mu[1, 2, 0, 5, 2].item(), atol=1e-6)


def test_tf_resize_is_centre_tap_average():
    # TF2 half-pixel bilinear at exact 2x/4x/8x = mean of the 2 centre taps per axis
    g = torch.Generator().manual_seed(4)
    img = torch.randn((1, 16, 32, 3), generator=g)
    for s in (2, 4, 8):
        out = rs.tf_resize_bilinear(img, (16 // s, 32 // s))
        a = s // 2 - 1
        ref = 0.25 * (img[:, a::s, a::s] + img[:, a::s, a + 1::s] + img[:, a + 1::s, a::s] + img[:, a + 1::s, a + 1::s])
        assert np.allclose(out, ref, atol=1e-6)


def test_loss_hand_cases():
    g = torch.Generator().manual_seed(5)
    tgt = torch.rand((2, 12, 20, 3), generator=g) * 2 - 1
    same = tgt.unsqueeze(1).repeat(1, 4, 1, 1, 1)
    # SSIM / L1 of identical images = 0
    assert torch.allclose(ref_loss.photometric_loss_ssim(same, tgt), torch.zeros(2), atol=1e-6)
    assert torch.allclose(ref_loss.photometric_loss_l1(same, tgt), torch.zeros(2), atol=1e-7)
    # L1 / SSIM of an all-invalid (black) warp = 0 (gray == 0 mask, loss_util.py:15-22)
    black = torch.zeros_like(same)
    assert torch.all(ref_loss.photometric_loss_l1(black, tgt) == 0)
    assert torch.all(ref_loss.photometric_loss_ssim(black, tgt) == 0)
    # smoothness of a constant disparity = 0
    disp = torch.full((2, 12, 20, 1), 0.3)
    assert torch.all(ref_loss.smootheness_loss(disp, tgt) == 0)
    # masked pixels stay in the mean's denominator: half-black warp gives half the L1
    half = same.clone()
    half[:, :, :, 10:] = 0
    other = torch.rand((2, 12, 20, 3), generator=g) * 2 - 1
    full_l1 = ref_loss.photometric_loss_l1(same, other, reduce=False)
    half_l1 = ref_loss.photometric_loss_l1(half, other)
    assert torch.allclose(half_l1, full_l1[:, :, :, :10].sum(dim=[1, 2, 3, 4]) / (4 * 12 * 20 * 3), atol=1e-6)


def test_oracle_gradcheck_fp64():
    # fp64 gradcheck of the restatement (no gradient golden values exist in the reference)
    g = torch.Generator().manual_seed(6)
    B, N, H, W = 1, 2, 8, 12
    src = torch.rand((B, N, H, W, 3), generator=g, dtype=torch.float64)
    K = torch.tensor([[[10., 0, 6], [0, 10., 4], [0, 0, 1]]], dtype=torch.float64)
    depth = (torch.rand((B, H, W, 1), generator=g, dtype=torch.float64) * 3 + 2).requires_grad_(True)
    pose = (torch.randn((B, N, 6), generator=g, dtype=torch.float64) * 0.02).requires_grad_(True)
    tgt = torch.rand((B, H, W, 3), generator=g, dtype=torch.float64)

    def fn(d, p):
        synth = rs.synthesize_multi_scale(src, K, [d], p)[0]
        return ref_loss.photometric_loss_l2(synth, tgt) + ref_loss.photometric_loss_ssim(synth, tgt)

    assert torch.autograd.gradcheck(fn, (depth, pose), eps=1e-7, atol=1e-5, rtol=1e-3, nondet_tol=0.0)


def test_rigid_inverse_matches_general_inverse():
    """utils/convert_pose.rigid_inverse (closed form, capturable) == tf.linalg.inv on poses (losses.py:90, 229)."""
    import torch
    from oracle import ref_pose
    from xpt_mde_2021_amd.utils import convert_pose as cp
    T = ref_pose.pose_rvec2matr_batch(torch.rand(3, 2, 6, dtype=torch.float64) * 2 - 1)
    inv = cp.rigid_inverse(T)
    assert torch.allclose(inv, torch.linalg.inv(T), atol=1e-12)
    assert torch.allclose(inv @ T, torch.eye(4, dtype=torch.float64).expand_as(T), atol=1e-12)
