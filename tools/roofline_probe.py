#!/usr/bin/env python3
"""Launches the fused warp+L1+SSIM kernels a few times at the bench shape (B=8) and at B=128 -- a small target for
`rocprofv3 --kernel-trace --stats` / `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes (profiles/README.md)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from xpt_mde_2021_amd.hip import ops, roofline as rf  # noqa: E402
from xpt_mde_2021_amd.utils import synthetic_data as sd  # noqa: E402

feats = {k: v.cuda() for k, v in sd.make_features(8, 128, 416).items()}
for batch in (8, 128):
    f, b, fb, bb, shape = rf.measure_fused(ops, feats, 10, batch=batch)
    print(f"B={batch}: fwd {f * 1e3:.1f} us ({fb / f / 1e6:.0f} GB/s algorithmic, {fb} B)  "
          f"bwd {b * 1e3:.1f} us ({bb / b / 1e6:.0f} GB/s algorithmic, {bb} B)")

for batch in (8, 128):
    f, b, fb, bb, shape = rf.measure_fused_ms(ops, feats, 10, batch=batch)
    print(f"B={batch} 4 scales in one launch: fwd {f * 1e3:.1f} us ({fb / f / 1e6:.0f} GB/s algorithmic, {fb} B)  "
          f"bwd {b * 1e3:.1f} us ({bb / b / 1e6:.0f} GB/s algorithmic, {bb} B)")

# calibration launches for the PMC passes: known byte counts with this kernel family's access widths
#   affine_act_fwd_kernel<float>: 4 B / lane loads and stores, reads 4n + writes 4n bytes
#   torch float4 copy (vectorized_elementwise / copyBuffer): 16 B / lane
n_rows, C = 1 << 20, 64                       # 256 MiB per tensor: beyond the Infinity Cache
x = torch.randn(1, C, n_rows // 64, 64, device="cuda").contiguous(memory_format=torch.channels_last)
bias = torch.zeros(C, device="cuda")
for _ in range(3):
    y = ops.bias_act(x, bias, 1.0)
    z = x.clone()
torch.cuda.synchronize()
print(f"calibration: affine_act_fwd_kernel<float> reads {x.numel() * 4} B and writes {x.numel() * 4} B per launch")
