"""Known-answer tests of the augmenters, restating the reference's own checks
(model/model_util/augmentation.py:227-330) plus crop/flip/colour behaviour on small CPU tensors."""
import numpy as np
import torch

from oracle import ref_pose
from xpt_mde_2021_amd.model.model_util import augmentation as aug


def test_random_crop_boxes():
    # augmentation.py:227-234
    cropper = aug.CropAndResize()
    for _ in range(20):
        boxes = cropper.random_crop_boxes(4)
        assert boxes.shape == (4, 4) and (boxes[0] == boxes[3]).all()
        wh = boxes[:, 2:] - boxes[:, :2]
        assert (wh.numpy() > 1 - cropper.half_crop_ratio * 2 - 1e-6).all()
        assert (boxes >= 0).all() and (boxes <= 1).all()


def test_adjust_intrinsic():
    # augmentation.py:237-259
    batch, height, width = 3, 200, 240
    intrinsic = torch.tensor([[[width / 2, 0, width / 2], [0, height / 2, height / 2], [0, 0, 1]]]).repeat(batch, 1, 1)
    xcrop, ycrop = 0.05, 0.1
    boxes = torch.tensor([[ycrop, xcrop, 1 - ycrop, 1 - xcrop]]).repeat(batch, 1)
    adj = aug.CropAndResize().adjust_intrinsic(intrinsic, boxes, (height, width)).numpy()
    assert np.isclose(adj[0], adj[-1]).all()
    assert np.isclose(adj[0, 0, 0], width / 2 / (1 - 2 * xcrop))
    assert np.isclose(adj[0, 0, 2], width / 2)
    assert np.isclose(adj[0, 1, 1], height / 2 / (1 - 2 * ycrop))
    assert np.isclose(adj[0, 1, 2], height / 2)
    assert np.isclose(adj[0, 2], [0, 0, 1]).all()


def test_flip_pose():
    # augmentation.py:285-304: flipping x negates tx, ry, rz of the twist
    torch.manual_seed(0)
    pose_vec = torch.rand(2, 2, 6) * 2 - 1
    pose_mat = ref_pose.pose_rvec2matr_batch(pose_vec)
    flipped = aug.HorizontalFlip().flip_gt_pose(pose_mat)
    vec_flip = ref_pose.pose_matr2rvec_batch(flipped)
    flip_vec = torch.tensor([-1., 1, 1, 1, -1, -1])
    assert np.isclose(pose_vec.numpy(), (vec_flip * flip_vec).numpy(), atol=1e-3).all()
    stereo = aug.HorizontalFlip().flip_stereo_pose(pose_mat[:, 0])
    assert torch.equal(stereo, flipped[:, 0])


def test_flip_intrinsic():
    # augmentation.py:307-330
    batch, height, width = 3, 200, 240
    intrinsic = torch.rand(batch, 3, 3) * 100 + 100
    flip = aug.HorizontalFlip().flip_intrinsic(intrinsic, (batch, height, width, 3)).numpy()
    k = intrinsic.numpy()
    assert np.isclose(k[:, 1:], flip[:, 1:]).all()
    assert np.isclose(k[:, 0, :2], flip[:, 0, :2]).all()
    assert np.isclose(width - k[:, 0, 2], flip[:, 0, 2]).all()


def _features(batch=2, snippet=5, h=16, w=24, stereo=True):
    g = torch.Generator().manual_seed(1)
    f = {"image5d": torch.rand(batch, snippet, h, w, 3, generator=g) * 2 - 1,
         "intrinsic": torch.tensor([[[w / 2., 0, w / 2.], [0, h / 2., h / 2.], [0, 0, 1]]]).repeat(batch, 1, 1),
         "depth_gt": torch.rand(batch, h, w, 1, generator=g) * 10,
         "pose_gt": ref_pose.pose_rvec2matr_batch(torch.rand(batch, 4, 6, generator=g) - 0.5)}
    if stereo:
        f["image5d_R"] = torch.rand(batch, snippet, h, w, 3, generator=g) * 2 - 1
        f["intrinsic_R"] = f["intrinsic"].clone()
        f["stereo_T_LR"] = ref_pose.pose_rvec2matr_batch(torch.rand(batch, 1, 6, generator=g) - 0.5)[:, 0]
    return f


def test_horizontal_flip_always_and_never():
    feats = _features()
    out = aug.TotalAugment([aug.HorizontalFlip(1.1)])(feats)
    assert torch.equal(out["image5d"], torch.flip(feats["image5d"], dims=[3]))
    assert torch.equal(out["image5d_R"], torch.flip(feats["image5d_R"], dims=[3]))
    assert np.isclose(out["intrinsic"][:, 0, 2].numpy(), 24 - 12.).all()
    assert torch.equal(out["depth_gt"], feats["depth_gt"])          # the reference leaves depth_gt unflipped
    assert not torch.equal(out["pose_gt"], feats["pose_gt"])
    same = aug.TotalAugment([aug.HorizontalFlip(-1.)])(feats)
    for k in feats:
        assert torch.equal(same[k], feats[k]), k
    # flipping twice is the identity
    twice = aug.TotalAugment([aug.HorizontalFlip(1.1), aug.HorizontalFlip(1.1)])(feats)
    for k in feats:
        assert torch.allclose(twice[k], feats[k], atol=1e-6), k


def test_crop_and_resize_identity_box_and_shift():
    img = torch.rand(3, 10, 12, 3)
    same = aug.crop_and_resize(img, torch.tensor([0., 0., 1., 1.]), (10, 12))
    assert torch.allclose(same, img, atol=1e-5)
    # a box one pixel in from every side, resized to the cropped size, is the plain slice
    box = torch.tensor([1 / 9., 1 / 11., 8 / 9., 10 / 11.])
    crop = aug.crop_and_resize(img, box, (8, 10))
    assert torch.allclose(crop, img[:, 1:9, 1:11], atol=1e-5)
    near = aug.crop_and_resize(img, box, (8, 10), method="nearest")
    assert torch.allclose(near, img[:, 1:9, 1:11], atol=1e-6)


def test_crop_keeps_feature_contract():
    feats = _features()
    cropper = aug.CropAndResize(0.3)
    out = aug.TotalAugment([cropper])(feats)
    for k in feats:
        assert out[k].shape == feats[k].shape, k
    box = cropper.param
    # the principal point moves with the crop and the focal length scales with it
    fx = feats["intrinsic"][0, 0, 0] / (box[3] - box[1])
    assert torch.isclose(out["intrinsic"][0, 0, 0], fx)
    assert torch.equal(feats["intrinsic"][0], _features()["intrinsic"][0])     # caller's dict untouched


def test_color_jitter():
    feats = _features(stereo=False)
    jit = aug.ColorJitter(1.1)
    out = aug.TotalAugment([jit])(feats)
    gamma, sat = jit.param
    assert 0.5 <= gamma <= 1.5 and 0.5 <= sat <= 1.5
    assert out["image5d"].min() >= -1 - 1e-5 and out["image5d"].max() <= 1 + 1e-5
    # restated per pixel: HSV saturation scale then gamma
    rgb = (feats["image5d"] + 1) / 2
    v, mn = rgb.max(-1, keepdim=True).values, rgb.min(-1, keepdim=True).values
    s = (v - mn) / v
    s2 = (s * sat).clamp(0, 1)
    expect = (v - (v - rgb) * s2 / s).clamp_min(0) ** gamma * 2 - 1
    assert torch.allclose(out["image5d"], expect, atol=1e-5)
    none = aug.TotalAugment([aug.ColorJitter(-1.)])(feats)
    assert torch.equal(none["image5d"], feats["image5d"])
    # grey pixels (zero saturation) only get the gamma
    grey = torch.full((1, 1, 2, 2, 3), 0.0)
    g = aug.ColorJitter(1.1)
    o = g({"image5d": grey.clone()})["image5d"]
    assert torch.allclose(o, 0.5 ** g.param[0] * 2 - 1)


def test_adjust_saturation_matches_hsv_roundtrip():
    import colorsys
    rng = np.random.default_rng(0)
    rgb = rng.random((50, 3)).astype(np.float32)
    for factor in (0.5, 1.0, 1.5):
        expect = []
        for r, g, b in rgb:
            h, s, v = colorsys.rgb_to_hsv(r, g, b)
            expect.append(colorsys.hsv_to_rgb(h, min(max(s * factor, 0), 1), v))
        got = aug.adjust_saturation(torch.from_numpy(rgb), torch.tensor(factor)).numpy()
        assert np.allclose(got, np.asarray(expect), atol=1e-5)


def test_augmentation_factory():
    a = aug.augmentation_factory({"CropAndResize": 0.2, "HorizontalFlip": 0.2, "ColorJitter": 0.2})
    assert [type(x).__name__ for x in a.augment_objects] == ["CropAndResize", "HorizontalFlip", "ColorJitter"]
    out = a(_features())
    assert out["image5d"].shape == (2, 5, 16, 24, 3)
    assert aug.augmentation_factory(None).augment_objects == []
    try:
        aug.augmentation_factory({"Rotate": 0.1})
        assert False
    except Exception as e:
        assert "Wrong augmentation type" in str(e)
