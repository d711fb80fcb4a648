#!/usr/bin/env python3
"""Launch time of the fused warp + L1 + SSIM kernels against the amount of work: batch 1..128 at 128x416 and the four
pyramid scales at batch 8.  The intercept of the line is the fixed cost of a launch (dispatch, first-row latency chain,
tail), the slope the per-pixel cost -- which of the two bounds the in-step shape (batch 8) decides what to optimise."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from xpt_mde_2021_amd.hip import ops, roofline as rf  # noqa: E402
from xpt_mde_2021_amd.utils import synthetic_data as sd  # noqa: E402

rows = []
for (B, H, W) in [(1, 128, 416), (2, 128, 416), (4, 128, 416), (8, 128, 416), (16, 128, 416), (32, 128, 416), (64, 128, 416),
                  (128, 128, 416), (8, 64, 208), (8, 32, 104), (8, 16, 52)]:
    feats = {k: v.cuda() for k, v in sd.make_features(min(B, 8), H, W).items()}
    f, b, fb, bb, shape = rf.measure_fused(ops, feats, 30, batch=B)
    rows.append((B, H, W, f * 1e3, b * 1e3, fb / f / 1e6, bb / b / 1e6))
    print(f"B={B:4d} {H}x{W}: fwd {f*1e3:7.1f} us ({fb/f/1e6:6.0f} GB/s)   bwd {b*1e3:7.1f} us ({bb/b/1e6:6.0f} GB/s)", flush=True)
