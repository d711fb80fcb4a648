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


def _sequential(feats, params, p_crop):
    """The augmenter chain composed from the per-class methods with the parameters the fused kernel reports."""
    box, flip, jit, gamma, sat = params[0:4], bool(params[4] > 0.5), bool(params[5] > 0.5), params[6], params[7]
    f = dict(feats)
    cropper = aug.CropAndResize(aug_prob=p_crop)
    cropper.random_crop_boxes = lambda n, device=None: box.unsqueeze(0).repeat(n, 1)
    f = cropper(f)
    if flip:
        f = aug.HorizontalFlip(aug_prob=1.1)(f)
    if jit:
        for key in ("image5d", "image5d_R"):
            if key in f:
                f[key] = aug.ColorJitter().jitter_color(f[key], gamma, sat)
    return f


@pytest.mark.parametrize("u", [
    [0.1, 0.2, 0.3, 0.4, 0.9, 0.9, 0.5, 0.5],          # identity box (all four draws clamp), no flip, no jitter
    [0.95, 0.99, 0.03, 0.01, 0.05, 0.9, 0.5, 0.5],     # all four sides cropped, flip
    [0.9, 0.1, 0.5, 0.02, 0.9, 0.1, 0.9, 0.1],         # two sides cropped, jitter (gamma 1.4, saturation 0.6)
    [0.97, 0.93, 0.06, 0.04, 0.1, 0.1, 0.05, 0.95],    # everything at once
])
@pytest.mark.parametrize("stereo", [False, True])
def test_fused_augmentation_equals_the_chain(gpu_device, u, stereo):
    """xpt_augment (one launch) == CropAndResize -> HorizontalFlip -> ColorJitter composed from the per-class methods, for
    images, ground-truth depth, intrinsics, ground-truth poses and the stereo extrinsic."""
    dev = gpu_device
    g = torch.Generator().manual_seed(3)
    B, S, H, W = 2, 5, 20, 36
    feats = {"image5d": (torch.rand(B, S, H, W, 3, generator=g) * 2 - 1).to(dev),
             "intrinsic": torch.tensor([[[18., 0, 17.5], [0, 16., 9.5], [0, 0, 1]]]).repeat(B, 1, 1).to(dev),
             "depth_gt": (torch.rand(B, H, W, 1, generator=g) * 10).to(dev),
             "pose_gt": ref_pose.pose_rvec2matr_batch(torch.rand(B, 4, 6, generator=g) - 0.5).to(dev)}
    if stereo:
        feats["image5d_R"] = (torch.rand(B, S, H, W, 3, generator=g) * 2 - 1).to(dev)
        feats["intrinsic_R"] = feats["intrinsic"] * 1.01
        feats["pose_gt_R"] = ref_pose.pose_rvec2matr_batch(torch.rand(B, 4, 6, generator=g) - 0.5).to(dev)
        feats["stereo_T_LR"] = ref_pose.pose_rvec2matr_batch(torch.rand(B, 1, 6, generator=g) - 0.5)[:, 0].to(dev)
    probs = {"CropAndResize": 0.2, "HorizontalFlip": 0.2, "ColorJitter": 0.2}
    total = aug.augmentation_factory(probs)
    assert total._fusable(feats)
    got = total._fused(feats, torch.tensor(u, device=dev))
    params = total.params
    assert bool(params[4] > 0.5) == (u[4] < 0.2) and bool(params[5] > 0.5) == (u[5] < 0.2)
    assert torch.allclose(params[6:8].cpu(), torch.tensor([u[6] + 0.5, u[7] + 0.5]))
    want = _sequential(feats, params, 0.2)
    assert set(got) == set(want) == set(feats)
    for key in want:
        a, b = got[key], want[key]
        assert a.shape == b.shape, key
        if key == "depth_gt":      # nearest neighbour: a sample within rounding of x.5 may pick the other texel
            assert (a != b).float().mean().item() < 2e-3, key
        elif key.startswith("image5d") and bool(params[5] > 0.5):
            # x ** gamma with gamma < 1 has an unbounded slope at 0: rounding-level differences of the bilinear sample
            # (1e-7) grow to 1e-4 on near-black pixels -- in any two implementations
            # -> compared before the gamma curve (x = ((y + 1) / 2) ** (1 / gamma)) at the usual bar, after it loosely
            inv = 1.0 / float(params[6])
            ua, ub = ((a + 1) / 2).clamp_min(0) ** inv, ((b + 1) / 2).clamp_min(0) ** inv
            assert torch.allclose(ua, ub, atol=3e-5, rtol=1e-5), (key, (ua - ub).abs().max().item())
            assert (a - b).abs().max().item() < 3e-3, (key, (a - b).abs().max().item())
        else:
            assert torch.allclose(a, b, atol=3e-5, rtol=1e-5), (key, (a - b).abs().max().item())
    # the inputs are untouched
    assert torch.equal(feats["intrinsic"][0], torch.tensor([[18., 0, 17.5], [0, 16., 9.5], [0, 0, 1]], device=dev))


def test_fused_augmentation_draws_like_the_chain(gpu_device):
    """The crop box the kernel derives from its uniforms has the distribution of CropAndResize.random_crop_boxes: each side
    is cropped with probability aug_prob, by at most 10 %; flip / jitter fire with their probabilities."""
    dev = gpu_device
    torch.manual_seed(11)
    feats = {"image5d": torch.zeros(1, 5, 8, 12, 3, device=dev), "intrinsic": torch.eye(3, device=dev).unsqueeze(0)}
    total = aug.augmentation_factory({"CropAndResize": 0.3, "HorizontalFlip": 0.25, "ColorJitter": 0.4})
    rows = []
    for _ in range(400):
        total(feats)
        rows.append(total.params.clone())
    p = torch.stack(rows).cpu()
    assert (p[:, 0:2] >= 0).all() and (p[:, 0:2] <= 0.1 + 1e-6).all() and (p[:, 2:4] >= 0.9 - 1e-6).all() and (p[:, 2:4] <= 1).all()
    cropped = torch.cat([(p[:, 0:2] > 0).float(), (p[:, 2:4] < 1).float()], dim=1).mean().item()
    assert abs(cropped - 0.3) < 0.05, cropped
    assert abs(p[:, 4].mean().item() - 0.25) < 0.07 and abs(p[:, 5].mean().item() - 0.4) < 0.08
    assert (p[:, 6:8] >= 0.5).all() and (p[:, 6:8] < 1.5).all()


def test_pinned_draws_repeat_inside_a_captured_graph_and_release(gpu_device):
    """TotalAugment.pin_draws / unpin_draws (the replay check of a captured training step, train_val._StepGraph): while
    pinned, every replay of ONE captured graph applies the same crop / flip / jitter; unpinned, the draws are fresh again --
    without re-capturing (the kernel reads the pin flag on the device)."""
    dev = gpu_device
    torch.manual_seed(5)
    g = torch.Generator().manual_seed(3)
    feats = {"image5d": torch.rand(2, 5, 16, 24, 3, generator=g).to(dev) * 2 - 1,
             "intrinsic": torch.tensor([[20., 0, 12], [0, 20., 8], [0, 0, 1]], device=dev).repeat(2, 1, 1)}
    total = aug.augmentation_factory({"CropAndResize": 0.9, "HorizontalFlip": 0.5, "ColorJitter": 0.9})
    assert total.can_pin()
    total(feats)                                   # eager warm-up: allocates the pin buffer
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = total(feats)
    params = total.params                          # the captured launch's parameter output (later calls rebind total.params)
    def replay():
        graph.replay()
        torch.cuda.synchronize()
        return out["image5d"].clone(), params.clone()
    total.pin_draws()
    a, pa = replay()
    b, pb = replay()
    eager = total(feats)                           # the eager path sees the pinned draws too
    torch.cuda.synchronize()
    assert torch.equal(a, b) and torch.equal(pa, pb) and torch.equal(eager["image5d"], a)
    total.unpin_draws()
    seen = {tuple(replay()[1].tolist()) for _ in range(6)}
    assert len(seen) > 1                           # fresh draws again
    assert tuple(pa.tolist()) not in seen
