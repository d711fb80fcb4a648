"""Would the data gradient and the weight gradient of a dense convolution overlap if they shared the chip?  For decoder /
PoseNet layer shapes: R launches of each kernel back to back on one stream (sum of the two) against the same launches
issued on two streams at once (no dependencies between them)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xpt_mde_2021_amd.hip import conv as C, lib as L, ops

dev = torch.device("cuda:0")
lib = L.load()
R = 40
shapes = [  # (B, cin, cout, H, W, k, stride, upsample)   decoder / PoseNet layers at batch 8, 128x416
    (8, 1056, 256, 4, 13, 3, 1, True), (8, 432, 256, 8, 26, 3, 1, False), (8, 256, 128, 8, 26, 3, 1, True),
    (8, 216, 128, 16, 52, 3, 1, False), (8, 128, 64, 16, 52, 3, 1, True), (8, 153, 64, 32, 104, 3, 1, False),
    (8, 64, 32, 32, 104, 3, 1, True), (8, 65, 32, 64, 208, 3, 1, False), (8, 32, 16, 64, 208, 3, 1, True),
    (8, 17, 16, 128, 416, 3, 1, False), (8, 32, 64, 32, 104, 3, 2, False), (8, 64, 128, 16, 52, 3, 2, False)]
tot = [0.0, 0.0, 0.0]
for (B, cin, cout, H, W, k, stride, ups) in shapes:
    cp = C.round_up(cin, 8)
    x = torch.randn((B, cp, H, W), device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    if cp != cin:
        x[:, cin:] = 0
    w = torch.randn((cout, cin, k, k), device=dev) * 0.05
    w.requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    y = C.conv2d_same(xr, w, None, stride, 1.0, ups)
    ctx = y.grad_fn
    g = torch.randn_like(y).contiguous(memory_format=torch.channels_last)
    Bq, PH, PW, Cp, Cc, xpitch, N, KH, KW, st, pt, pl, OH, OW, u, slope = ctx.geom
    e = C.packer.get(w, need_bwd=True)
    dx = torch.empty((B, Cp, PH, PW), dtype=torch.bfloat16, device=dev, memory_format=torch.channels_last)
    xs = ctx.saved_tensors[0]

    def dgrad(stream):
        L.check(lib.xpt_conv2d_bwd_data(g.data_ptr(), e["bwd"].data_ptr(), dx.data_ptr(), B, OH, OW, e["Np"], N, Cp, KH, KW,
                                        st, pt, pl, PH, PW, Cp, u, stream.cuda_stream), "dgrad")

    nsplit = lib.xpt_conv2d_bwd_weight_splits(B, Cp, N, KH, KW, st, OH, OW)
    ws = torch.empty(nsplit * N * KH * KW * Cc, dtype=torch.float32, device=dev)

    def wgrad(stream):
        L.check(lib.xpt_conv2d_bwd_weight_partials(g.data_ptr(), xs.data_ptr(), ws.data_ptr(), ws.numel(), B, PH, PW, Cp, Cc,
                                                   xpitch, N, N, KH, KW, st, pt, pl, OH, OW, u, stream.cuda_stream), "wgrad")

    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def timed(fn):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        torch.cuda.synchronize()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) * 1e3 / R

    def both():
        s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
        for _ in range(R):
            dgrad(s1); wgrad(s2)

    cur = torch.cuda.current_stream()
    for _ in range(3): dgrad(cur); wgrad(cur)
    td = timed(lambda: [dgrad(cur) for _ in range(R)])
    tw = timed(lambda: [wgrad(cur) for _ in range(R)])
    tb = timed(both)
    tot[0] += td; tot[1] += tw; tot[2] += tb
    print(f"[pair] cin {cin:4d} cout {cout:3d} {H:3d}x{W:3d} s{stride} up{int(ups)}: dgrad {td:6.1f} us  wgrad {tw:6.1f} us  "
          f"sum {td + tw:6.1f}  two streams {tb:6.1f}", flush=True)
print(f"[pair] total: dgrad {tot[0]:.0f} wgrad {tot[1]:.0f} sum {tot[0] + tot[1]:.0f} two streams {tot[2]:.0f} us")
