"""Per-layer timing of the dense convolutions at the bench shape (batch 8, 128x416): matrix-core kernels of this repo
(forward / data gradient / weight-gradient partials) next to MIOpen's forward / backward for the same shape.

    python tools/bench_conv.py [batch] [plan ...]      plan = xpt_conv2d_tune code (RM*100 + RN*10 + log2 NKW), 0 = auto
"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from xpt_mde_2021_amd.hip import conv as xc, lib as _lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
plans = [int(a) for a in sys.argv[2:]] or [0]
lib = _lib.load()
LAYERS = [
    ("pose0", 15, 32, 5, 2, 128, 416, False), ("pose1", 32, 32, 5, 2, 64, 208, False),
    ("pose2", 32, 64, 3, 2, 32, 104, False), ("pose3", 64, 128, 3, 2, 16, 52, False),
    ("pose4", 128, 256, 3, 2, 8, 26, False), ("pose5", 256, 256, 3, 2, 4, 13, False),
    ("pose6", 256, 256, 3, 1, 2, 7, False), ("pose7", 256, 256, 3, 1, 2, 7, False),
    ("up4a", 1056, 256, 3, 1, 4, 13, True), ("up4b", 432, 256, 3, 1, 8, 26, False),
    ("up3a", 256, 128, 3, 1, 8, 26, True), ("up3b", 216, 128, 3, 1, 16, 52, False),
    ("up2a", 128, 64, 3, 1, 16, 52, True), ("up2b", 87, 64, 3, 1, 32, 104, False),
    ("up1a", 64, 32, 3, 1, 32, 104, True), ("up1b", 65, 32, 3, 1, 64, 208, False),
    ("up0a", 32, 16, 3, 1, 64, 208, True), ("up0b", 17, 16, 3, 1, 128, 416, False),
]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


tot = {}
print(f"batch {B}; us per call: fwd / dgrad / wgrad(partials) per plan {plans}; MIOpen bf16 fwd / bwd(dx+dw)")
for name, cin, cout, k, s, H, W, ups in LAYERS:
    cp = xc.round_up(cin, 8)
    x = torch.randn(B, cp, H, W, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, k, k, device="cuda") / math.sqrt(cin * k * k)).contiguous(memory_format=torch.channels_last)
    bias = torch.zeros(cout, device="cuda")
    e = xc.packer.get(w, need_bwd=True)
    Hl, Wl = H << ups, W << ups
    (pt, pb), (pl, pr) = xc.same_pad(Hl, k, s), xc.same_pad(Wl, k, s)
    OH, OW = -(-Hl // s), -(-Wl // s)
    y = torch.empty((B, cout, OH, OW), dtype=torch.bfloat16, device="cuda", memory_format=torch.channels_last)
    g = torch.randn_like(y)
    dx = torch.empty_like(x)
    st = torch.cuda.current_stream().cuda_stream
    nsplit = lib.xpt_conv2d_bwd_weight_splits(B, cp, cout, k, k, s, OH, OW)
    part = torch.empty(nsplit * cout * k * k * cin, dtype=torch.float32, device="cuda")
    flops = 2.0 * B * OH * OW * cout * k * k * cin
    row = []
    for plan in plans:
        lib.xpt_conv2d_tune(-1000 if plan == -1 else -1512)      # plan -1: automatic WITHOUT the LDS-staged kernel
        lib.xpt_conv2d_tune(max(plan, 0))
        t_f = timeit(lambda: _lib.check(lib.xpt_conv2d_fwd(x.data_ptr(), e["fwd"].data_ptr(), bias.data_ptr(), y.data_ptr(), B, H, W,
                                                           cp, cp, cout, k, k, s, pt, pl, OH, OW, cout, int(ups), 0.1,
                                                           torch.cuda.current_stream().cuda_stream), "fwd"))
        t_d = timeit(lambda: _lib.check(lib.xpt_conv2d_bwd_data(g.data_ptr(), e["bwd"].data_ptr(), dx.data_ptr(), B, OH, OW, e["Np"],
                                                                cout, cp, k, k, s, pt, pl, H, W, cp, int(ups),
                                                                torch.cuda.current_stream().cuda_stream), "dgrad"))
        row.append((t_f, t_d))
        tot[plan] = tot.get(plan, 0.0) + t_f + t_d
    lib.xpt_conv2d_tune(-1512)
    lib.xpt_conv2d_tune(0)
    t_w = timeit(lambda: _lib.check(lib.xpt_conv2d_bwd_weight_partials(g.data_ptr(), x.data_ptr(), part.data_ptr(), part.numel(), B, H, W,
                                                                       cp, cin, cp, cout, cout, k, k, s, pt, pl, OH, OW, int(ups),
                                                                       torch.cuda.current_stream().cuda_stream), "wgrad"))
    tot["w"] = tot.get("w", 0.0) + t_w
    # library: pad + (upsample) + conv, bf16
    torch.backends.cudnn.benchmark = True
    xin = F.interpolate(x, scale_factor=2, mode="nearest") if ups else x
    xin = F.pad(xin, (pl, pr, pt, pb)).contiguous(memory_format=torch.channels_last)
    wl = F.pad(w, (0, 0, 0, 0, 0, cp - cin)).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    t_lf = timeit(lambda: F.conv2d(xin, wl, None, s))
    t_lb = timeit(lambda: torch.ops.aten.convolution_backward(g, xin, wl, None, [s, s], [0, 0], [1, 1], False, [0, 0], 1,
                                                              [True, True, False]))
    tot["lf"] = tot.get("lf", 0.0) + t_lf
    tot["lb"] = tot.get("lb", 0.0) + t_lb
    cells = " | ".join(f"{a:6.1f} {b:6.1f}" for a, b in row)
    print(f"{name:6s} {cin:4d}->{cout:3d} k{k} s{s} {OH:3d}x{OW:3d} splits {nsplit:3d} | {cells} | w {t_w:6.1f} | lib {t_lf:6.1f} {t_lb:6.1f} | "
          f"fwd {flops / row[0][0] * 1e-6:6.1f} TF/s", flush=True)
print("totals us:", {str(k): round(v, 1) for k, v in tot.items()})
