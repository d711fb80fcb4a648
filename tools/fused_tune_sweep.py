#!/usr/bin/env python3
"""Rows-per-wave sweep of the fused warp + L1 + SSIM kernels at the in-step shapes (batch 8, four scales) and at batch 128."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from xpt_mde_2021_amd.hip import ops, roofline as rf, lib as _lib  # noqa: E402
from xpt_mde_2021_amd.utils import synthetic_data as sd  # noqa: E402

lib = _lib.load()
shapes = [(8, 128, 416), (8, 64, 208), (8, 32, 104), (8, 16, 52), (128, 128, 416)]
feats = {hw: {k: v.cuda() for k, v in sd.make_features(8, hw[0], hw[1]).items()} for hw in {(h, w) for _, h, w in shapes}}
for fw, bw, mr in ((4096, 1536, 8), (8192, 3072, 4), (8192, 4096, 4), (16384, 6144, 2), (16384, 16384, 2), (16384, 3072, 2)):
    assert lib.xpt_photo_fused_tune(fw, bw, mr) == 0
    out = []
    for B, H, W in shapes:
        f, b, fb, bb, shape = rf.measure_fused(ops, feats[(H, W)], 30, batch=B)
        out.append(f"B{B} {H}x{W}: {f*1e3:6.1f} / {b*1e3:6.1f}")
    print(f"fwd_min_waves {fw:6d} bwd_min_waves {bw:6d} min_rows {mr} | us fwd / bwd | " + " | ".join(out), flush=True)
