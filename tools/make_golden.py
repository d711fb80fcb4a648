#!/usr/bin/env python3
"""Writes tests/golden/*.npz: seeded inputs and the ORACLE's outputs for the big parity cases of the hot path.

The reference (TensorFlow 2.4) cannot be executed in this pipeline, so these vectors are NOT reference outputs: they
freeze the CPU restatement (oracle/, fp64 arithmetic, stored as fp32) so that drift of the oracle between rounds is
caught (tests/test_golden.py, CPU) and the HIP kernels are compared with fixed numbers (GPU).  Re-run only when the oracle
is deliberately changed:   python tools/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_loss, ref_pose, ref_synthesize as rs  # noqa: E402
from xpt_mde_2021_amd.utils import synthetic_data as sd  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def synthesis_case(seed, B, N, h, w, name, with_maps=True):
    g = torch.Generator().manual_seed(seed)
    src = torch.stack([sd.smooth_noise((B, h, w, 3), g) for _ in range(N)], dim=1)
    tgt = sd.smooth_noise((B, h, w, 3), g)
    depth = sd.smooth_depth(B, h, w, g, lo=4.0, hi=40.0)
    K = sd.kitti_like_intrinsic(B, h, w)
    pose = sd.random_poses(B, N, g) * 0.3
    # flip-aware fixtures (tests/util.py): the depth of pixels whose projection lies within fp32 rounding of an integer
    # coordinate or of the validity border is moved by a few per cent until it no longer does, so that the GPU comparison
    # needs no allowance for floor() / validity flips
    from tests.util import flip_safe_depth
    depth, _ = flip_safe_depth(depth, ref_pose.pose_rvec2matr_batch(pose.double()), K, 1, nudge=True)
    d64 = depth.double().requires_grad_(True)
    p64 = pose.double().requires_grad_(True)
    synth = rs.synthesize_multi_scale(src.double(), K.double(), [d64], p64)[0]
    l1_map = ref_loss.photometric_loss_l1(synth, tgt.double(), False)
    ss_map = ref_loss.photometric_loss_ssim(synth, tgt.double(), False)
    l1 = ref_loss.photometric_loss_l1(synth, tgt.double())
    ss = ref_loss.photometric_loss_ssim(synth, tgt.double())
    (l1.sum() + ss.sum()).backward()
    disp = ref_loss.safe_reciprocal_number(depth.double()).requires_grad_(True)
    smooth = ref_loss.smootheness_loss(disp, tgt.double())
    smooth.sum().backward()
    maps = {"l1_map": l1_map.detach().float().numpy(), "ssim_map": ss_map.detach().float().numpy()} if with_maps else {}
    np.savez_compressed(os.path.join(OUT, name), src=src.numpy(), target=tgt.numpy(), depth=depth.numpy(), intrinsic=K.numpy(),
                        pose=pose.numpy(), synth=synth.detach().float().numpy(), **maps, l1=l1.detach().float().numpy(),
                        ssim=ss.detach().float().numpy(), d_depth=d64.grad.float().numpy(), d_pose=p64.grad.float().numpy(),
                        smooth=smooth.detach().float().numpy(), d_disp=disp.grad.float().numpy())


def pose_case(seed, name):
    g = torch.Generator().manual_seed(seed)
    pose = sd.random_poses(6, 4, g).double()
    mat = ref_pose.pose_rvec2matr_batch(pose)
    back = ref_pose.pose_matr2rvec_batch(mat)
    np.savez_compressed(os.path.join(OUT, name), pose=pose.float().numpy(), matrix=mat.float().numpy(), twist_back=back.float().numpy())


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    synthesis_case(20211119, 1, 2, 20, 32, "synth_loss_1x2x20x32.npz")
    synthesis_case(7, 1, 4, 12, 40, "synth_loss_1x4x12x40.npz", with_maps=False)
    pose_case(3, "pose_6x4.npz")
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)), "bytes")
