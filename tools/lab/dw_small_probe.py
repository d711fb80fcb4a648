"""Lab: multi-layer depthwise (small-map kernels, whole map of an image in LDS) against the single-layer kernel, element by element."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from xpt_mde_2021_amd.hip import ops
dev = torch.device("cuda:0")
for C, ks, H, W in ((176, [5, 7, 7, 5, 3], 10, 13), (88, [5, 3, 3, 5, 3], 10, 13), (176, [7, 7, 7, 7, 7], 4, 13), (44, [5, 3, 3, 5, 3], 16, 52)):
    g = torch.Generator().manual_seed(C)
    h = torch.randn(2, C, H, W, generator=g).to(dev, torch.bfloat16).contiguous(memory_format=torch.channels_last)
    p = torch.randn(2, C, H, W, generator=g).to(dev, torch.bfloat16).contiguous(memory_format=torch.channels_last)
    ws = [(torch.randn(C, 1, k, k, generator=g) * 0.2).to(dev) for k in ks]
    ins = [h, h, p, p, p]
    pads = [(k // 2,) * 4 for k in ks]
    with torch.no_grad():
        ya = ops.multi_depthwise(ins, ws, relu_in=True, stride=1, pads=pads)
        yb = [ops.depthwise_conv2d(x, q, 1, pd, True) for x, q, pd in zip(ins, ws, pads)]
    torch.cuda.synchronize()
    for j, (a, b) in enumerate(zip(ya, yb)):
        bad = (a != b)
        idx = torch.nonzero(bad)
        print(C, ks[j], H, W, "mismatches", int(bad.sum()), "of", a.numel(), idx[:6].tolist())
