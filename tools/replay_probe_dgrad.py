#!/usr/bin/env python3
"""Which library convolution backward of the rigid step breaks under hipGraph replay, layer by layer?

Each dense convolution shape of PoseNetImproved and of the depth decoder (batch 8, 128x416) gets ITS OWN hipGraph that
holds nothing but one `aten.convolution_backward` call (data gradient, weight gradient, or both) on fixed random
operands; the graph is replayed `R` times and every replay is compared with the eager result.  The first replay that
differs names the (layer, gradient kind, dtype) whose solver does not survive replay -- no code of this repo runs in the
graphs, so a hit is a library defect, a clean sheet points back at the repo's own persistent state.

    python tools/replay_probe_dgrad.py [bf16|fp32] [find|nofind] [replays]
"""
import sys

import torch
import torch.nn.functional as F

dtype = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else torch.float32
find = len(sys.argv) < 3 or sys.argv[2] == "find"
R = int(sys.argv[3]) if len(sys.argv) > 3 else 4
torch.backends.cudnn.benchmark = find
torch.manual_seed(0)
B = 8
# (name, cin, cout, k, stride, H_in, W_in, (pad_t, pad_b, pad_l, pad_r))  -- TF-SAME paddings as layer_ops.same_pad gives
LAYERS = [
    ("pose0", 15, 32, 5, 2, 128, 416, (1, 2, 1, 2)), ("pose1", 32, 32, 5, 2, 64, 208, (1, 2, 1, 2)),
    ("pose2", 32, 64, 3, 2, 32, 104, (0, 1, 0, 1)), ("pose3", 64, 128, 3, 2, 16, 52, (0, 1, 0, 1)),
    ("pose4", 128, 256, 3, 2, 8, 26, (0, 1, 0, 1)), ("pose5", 256, 256, 3, 2, 4, 13, (0, 1, 1, 1)),
    ("pose6", 256, 256, 3, 1, 2, 7, (1, 1, 1, 1)), ("pose7", 256, 256, 3, 1, 2, 7, (1, 1, 1, 1)),
    ("up4a", 1056, 256, 3, 1, 8, 26, (1, 1, 1, 1)), ("up4b", 432, 256, 3, 1, 8, 26, (1, 1, 1, 1)),
    ("up3a", 256, 128, 3, 1, 16, 52, (1, 1, 1, 1)), ("up3b", 216, 128, 3, 1, 16, 52, (1, 1, 1, 1)),
    ("up2a", 128, 64, 3, 1, 32, 104, (1, 1, 1, 1)), ("up2b", 87, 64, 3, 1, 32, 104, (1, 1, 1, 1)),
    ("up1a", 64, 32, 3, 1, 64, 208, (1, 1, 1, 1)), ("up1b", 65, 32, 3, 1, 64, 208, (1, 1, 1, 1)),
    ("up0a", 32, 16, 3, 1, 128, 416, (1, 1, 1, 1)), ("up0b", 17, 16, 3, 1, 128, 416, (1, 1, 1, 1)),
    ("head3", 128, 1, 3, 1, 16, 52, (1, 1, 1, 1)), ("head0", 16, 1, 3, 1, 128, 416, (1, 1, 1, 1)),
]


def rel(a, b):
    d = torch.nan_to_num((a.float() - b.float()).abs(), nan=3e38, posinf=3e38)
    return float(d.max()) / max(float(b.float().abs().max()), 1e-20)


def probe(name, cin, cout, k, s, H, W, pad, mask):
    pt, pb, pl, pr = pad
    x = torch.randn(B, cin, H + pt + pb, W + pl + pr, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
    w = (0.05 * torch.randn(cout, cin, k, k, device="cuda")).to(dtype).contiguous(memory_format=torch.channels_last)
    y = F.conv2d(x, w, None, s)
    dy = torch.randn_like(y).contiguous(memory_format=torch.channels_last)

    def run():
        return torch.ops.aten.convolution_backward(dy, x, w, None, [s, s], [0, 0], [1, 1], False, [0, 0], 1, mask)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            ref = run()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    ref = [t.clone() if t is not None else None for t in ref]
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = run()
    worst = []
    for it in range(R):
        for t in out:
            if t is not None:
                t.fill_(float("nan"))                 # a replay must overwrite every element
        g.replay()
        torch.cuda.synchronize()
        worst.append(max(rel(o, r) for o, r in zip(out, ref) if o is not None))
    tol = 3e-2 if dtype == torch.bfloat16 else 1e-3
    flag = "BAD" if max(worst) > tol else "ok"
    print(f"[probe] {name:6s} {['', 'dx'][mask[0]]}{['', 'dw'][mask[1]]:2s} {str(dtype)[6:]:8s} find={int(find)} "
          f"replay errors {' '.join(f'{e:.1e}' for e in worst)}  {flag}", flush=True)
    return flag == "BAD"


bad = 0
for layer in LAYERS:
    for mask in ([True, False, False], [False, True, False]):
        if layer[0] == "pose0" and mask[0]:
            continue                                    # the image needs no gradient
        try:
            bad += probe(*layer, mask)
        except Exception as e:                          # noqa: BLE001 -- a probe that cannot run is reported, not fatal
            print(f"[probe] {layer[0]} {mask}: {type(e).__name__}: {str(e)[:200]}", flush=True)
print(f"[probe] RESULT: {bad} defective (layer, gradient) pairs", flush=True)
