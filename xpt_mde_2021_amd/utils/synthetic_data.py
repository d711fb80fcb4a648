"""Seeded, KITTI-shaped synthetic inputs for benchmarks, smoke tests and parity tests.

Feature-dict contract follows tfrecords/tfrecord_reader.py:61-108 of the reference:
image5d [B,5,H,W,3] in [-1,1] with the TARGET FRAME LAST, intrinsic [B,3,3], depth_gt [B,H,W,1],
pose_gt [B,4,4,4]; stereo adds *_R and stereo_T_LR [B,4,4].
"""
import math

import torch
import torch.nn.functional as F


def smooth_noise(shape_bhwc, generator, cutoff=8, dtype=torch.float32):
    """Low-pass-filtered U(-1,1) noise [B,H,W,C]: random coarse grid, bicubic up-sampling, clamped to [-1,1]."""
    b, h, w, c = shape_bhwc
    gh, gw = max(h // cutoff, 2), max(w // cutoff, 2)
    coarse = torch.rand((b, c, gh, gw), generator=generator, dtype=dtype) * 2 - 1
    fine = F.interpolate(coarse, size=(h, w), mode="bicubic", align_corners=True)
    detail = (torch.rand((b, c, h, w), generator=generator, dtype=dtype) * 2 - 1) * 0.05
    return (fine * 0.9 + detail).clamp(-1, 1).permute(0, 2, 3, 1).contiguous()


def kitti_like_intrinsic(batch, height, width, dtype=torch.float32):
    """[[0.58 W, 0, 0.5 W], [0, 1.92 H, 0.5 H], [0, 0, 1]]  (fx~241, fy~246 at 128x416)."""
    K = torch.tensor([[0.58 * width, 0., 0.5 * width], [0., 1.92 * height, 0.5 * height], [0., 0., 1.]], dtype=dtype)
    return K.unsqueeze(0).repeat(batch, 1, 1).contiguous()


def random_poses(batch, numsrc, generator, dtype=torch.float32):
    """N(0, diag(0.3 m, 0.05 m, 1.0 m, 0.01, 0.02, 0.01 rad)); never exactly zero rotation."""
    std = torch.tensor([0.3, 0.05, 1.0, 0.01, 0.02, 0.01], dtype=dtype)
    p = torch.randn((batch, numsrc, 6), generator=generator, dtype=dtype) * std
    p[..., 3:] += 1e-4
    return p


def smooth_depth(batch, height, width, generator, lo=1.0, hi=80.0, dtype=torch.float32):
    d = smooth_noise((batch, height, width, 1), generator, cutoff=16, dtype=dtype)
    return ((d + 1) * 0.5 * (hi - lo) + lo).clamp(lo, hi).contiguous()


def make_features(batch, height=128, width=416, snippet=5, seed=20211119, stereo=False, dtype=torch.float32):
    """Synthetic feature dict: target = smooth texture, sources = target shifted by (+-2, +-1) px-scale motion."""
    g = torch.Generator().manual_seed(seed)
    feats = {}

    def one_side(sfx, shift0):
        target = smooth_noise((batch, height, width, 3), g, dtype=dtype)
        frames = []
        for i in range(snippet - 1):
            dx = (-2, -1, 1, 2)[i % 4] + shift0
            dy = (1, 0, 0, -1)[i % 4]
            frames.append(torch.roll(target, shifts=(dy, dx), dims=(1, 2)))
        frames.append(target)
        image5d = torch.stack(frames, dim=1).contiguous()
        feats["image5d" + sfx] = image5d
        feats["image" + sfx] = image5d.reshape(batch, snippet * height, width, 3)
        feats["intrinsic" + sfx] = kitti_like_intrinsic(batch, height, width, dtype)
        depth = smooth_depth(batch, height, width, g, dtype=dtype)
        lidar_mask = (torch.rand((batch, height, width, 1), generator=g) < 0.05).to(dtype)
        feats["depth_gt" + sfx] = depth * lidar_mask
        pose_gt = torch.eye(4, dtype=dtype).reshape(1, 1, 4, 4).repeat(batch, snippet - 1, 1, 1)
        pose_gt[:, :, 2, 3] = torch.tensor([-2., -1., 1., 2.][: snippet - 1], dtype=dtype) * 0.5
        feats["pose_gt" + sfx] = pose_gt

    one_side("", 0)
    if stereo:
        one_side("_R", 3)
        T = torch.eye(4, dtype=dtype).unsqueeze(0).repeat(batch, 1, 1)
        T[:, 0, 3] = 0.54
        feats["stereo_T_LR"] = T
    return feats


def tfr_config_for(feats):
    """The `tfr_config` dict the reference reads from tfr_config.txt (tfrecord_reader.py:31-45): key membership
    drives model / loss selection; `imshape` = [snippet, H, W, 3]."""
    cfg = {k: {"parse_type": "bytes", "decode_type": "float32", "shape": list(v.shape[1:])} for k, v in feats.items()}
    cfg["imshape"] = list(feats["image5d"].shape[1:])
    cfg["length"] = int(feats["image5d"].shape[0])
    return cfg
