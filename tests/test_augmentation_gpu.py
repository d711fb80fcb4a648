"""The reference's own augmentation known answers (model/model_util/augmentation.py:237-330: adjust_intrinsic, flip_pose,
flip_intrinsic) and the flip / crop image semantics evaluated ON THE DEVICE (the training step runs the augmenters on GPU
tensors inside the captured graph), against the same expected values the CPU tests use."""
import numpy as np
import pytest
import torch

from oracle import ref_pose
from xpt_mde_2021_amd.model.model_util import augmentation as aug

pytestmark = pytest.mark.gpu


def test_reference_known_answers_on_device(gpu_device):
    dev = gpu_device
    batch, height, width = 3, 200, 240
    intrinsic = torch.tensor([[[width / 2, 0, width / 2], [0, height / 2, height / 2], [0, 0, 1]]]).repeat(batch, 1, 1).to(dev)
    xcrop, ycrop = 0.05, 0.1
    boxes = torch.tensor([[ycrop, xcrop, 1 - ycrop, 1 - xcrop]]).repeat(batch, 1).to(dev)
    adj = aug.CropAndResize().adjust_intrinsic(intrinsic, boxes, (height, width)).cpu().numpy()        # :237-259
    assert np.isclose(adj[0, 0, 0], width / 2 / (1 - 2 * xcrop)) and np.isclose(adj[0, 0, 2], width / 2)
    assert np.isclose(adj[0, 1, 1], height / 2 / (1 - 2 * ycrop)) and np.isclose(adj[0, 1, 2], height / 2)
    assert np.isclose(adj[0, 2], [0, 0, 1]).all()
    torch.manual_seed(0)
    pose_vec = torch.rand(2, 2, 6) * 2 - 1
    flipped = aug.HorizontalFlip().flip_gt_pose(ref_pose.pose_rvec2matr_batch(pose_vec).to(dev)).cpu()    # :285-304
    vec_flip = ref_pose.pose_matr2rvec_batch(flipped)
    assert np.isclose(pose_vec.numpy(), (vec_flip * torch.tensor([-1., 1, 1, 1, -1, -1])).numpy(), atol=1e-3).all()
    k = torch.rand(batch, 3, 3) * 100 + 100
    flip = aug.HorizontalFlip().flip_intrinsic(k.to(dev), (batch, height, width, 3)).cpu().numpy()       # :307-330
    assert np.isclose(k.numpy()[:, 1:], flip[:, 1:]).all() and np.isclose(width - k.numpy()[:, 0, 2], flip[:, 0, 2]).all()


def test_flip_and_identity_crop_images_on_device(gpu_device):
    dev = gpu_device
    g = torch.Generator().manual_seed(1)
    feats = {"image5d": (torch.rand(2, 5, 16, 24, 3, generator=g) * 2 - 1).to(dev),
             "intrinsic": torch.tensor([[[12., 0, 12.], [0, 8., 8.], [0, 0, 1]]]).repeat(2, 1, 1).to(dev),
             "depth_gt": (torch.rand(2, 16, 24, 1, generator=g) * 10).to(dev),
             "pose_gt": ref_pose.pose_rvec2matr_batch(torch.rand(2, 4, 6, generator=g) - 0.5).to(dev)}
    out = aug.HorizontalFlip(aug_prob=1.1)(dict(feats))
    assert torch.equal(out["image5d"], feats["image5d"].flip(3))
    assert torch.equal(out["depth_gt"], feats["depth_gt"])          # the reference leaves depth_gt alone (augmentation.py:147-166)
    cropper = aug.CropAndResize(aug_prob=1.1)
    cropper.random_crop_boxes = lambda n, device=None: torch.tensor([[0., 0., 1., 1.]], device=device).repeat(n, 1)   # identity box
    same = cropper(dict(feats))
    assert torch.allclose(same["image5d"], feats["image5d"], atol=1e-5)
    assert torch.allclose(same["intrinsic"], feats["intrinsic"], atol=1e-4)
