"""Half- / full-resolution decoder convolutions (dp_up2 ... dp_up0, batch 8): the persistent weight-stationary kernels
(csrc/xpt_conv_stream.hip) against conv_halo_kernel (csrc/xpt_conv.hip), forward and data gradient, with the launch's
algorithmic bytes (input + output once, bf16) and its fraction of the 8 TB/s HBM roofline.

    python tools/bench_stream.py [batch]
"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from xpt_mde_2021_amd.hip import conv as xc, lib as _lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ONE = sys.argv[2] if len(sys.argv) > 2 else None           # PMC passes: this layer only, few launches (halo + stream at 3 workgroups per CU)
lib = _lib.load()
LAYERS = [
    ("up2a", 128, 64, 16, 52, True), ("up2b", 96, 64, 32, 104, False),
    ("up1a", 64, 32, 32, 104, True), ("up1b", 72, 32, 64, 208, False),
    ("up0a", 32, 16, 64, 208, True), ("up0b", 24, 16, 128, 416, False),
]


def timeit(fn, n=20):
    if ONE:
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        return 1.0
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best


print(f"batch {B}; us per call fwd/dgrad: halo kernel | stream 1 / 2 / 3 / 4 workgroups per CU | MB moved, best fraction of 8 TB/s")
for name, cin, cout, H, W, ups in LAYERS:
    if ONE and name != ONE:
        continue
    k, s = 3, 1
    cp = xc.round_up(cin, 8)
    x = torch.randn(B, cp, H, W, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, k, k, device="cuda") / math.sqrt(cin * k * k)).contiguous(memory_format=torch.channels_last)
    bias = torch.zeros(cout, device="cuda")
    e = xc.packer.get(w, need_bwd=True)
    Hl, Wl = H << ups, W << ups
    (pt, _), (pl, _) = xc.same_pad(Hl, k, s), xc.same_pad(Wl, k, s)
    OH, OW = Hl, Wl
    y = torch.empty((B, cout, OH, OW), dtype=torch.bfloat16, device="cuda", memory_format=torch.channels_last)
    y2 = torch.empty_like(y)
    g = torch.randn_like(y)
    dx = torch.empty_like(x)
    dx2 = torch.empty_like(x)
    mb = (x.numel() + y.numel()) * 2 / 1e6

    def old_f():
        _lib.check(lib.xpt_conv2d_fwd(x.data_ptr(), e["fwd"].data_ptr(), bias.data_ptr(), y.data_ptr(), B, H, W, cp, cp, cout, k, k, s,
                                      pt, pl, OH, OW, cout, int(ups), 0.1, torch.cuda.current_stream().cuda_stream), "fwd")

    def old_d():
        _lib.check(lib.xpt_conv2d_bwd_data(g.data_ptr(), e["bwd"].data_ptr(), dx.data_ptr(), B, OH, OW, e["Np"], cout, cp, k, k, s,
                                           pt, pl, H, W, cp, int(ups), torch.cuda.current_stream().cuda_stream), "dgrad")

    def new_f():
        _lib.check(lib.xpt_conv2d_fwd_stream(x.data_ptr(), e["fwd"].data_ptr(), bias.data_ptr(), y2.data_ptr(), B, H, W, cp, cp, cout,
                                             pt, pl, OH, OW, cout, int(ups), 0.1, torch.cuda.current_stream().cuda_stream), "fwd stream")

    def new_d():
        _lib.check(lib.xpt_conv2d_bwd_data_stream(g.data_ptr(), e["bwd"].data_ptr(), dx2.data_ptr(), B, OH, OW, e["Np"], cout, cp,
                                                  pt, pl, H, W, cp, int(ups), torch.cuda.current_stream().cuda_stream), "dgrad stream")

    t_of, t_od = timeit(old_f), timeit(old_d)
    cells, best_f, best_d = [], 1e9, 1e9
    ef = ed = float('nan')
    for wgs in ((3,) if ONE else (1, 2, 3, 4)):
        lib.xpt_conv2d_stream_tune(1, 1, wgs, 150)
        if not lib.xpt_conv2d_stream_serves(B, OH, OW, cout, cp, 3, 3, 1, int(ups)):
            cells.append("  -  /  -  ")
            continue
        t_f, t_d = timeit(new_f), timeit(new_d)
        best_f, best_d = min(best_f, t_f), min(best_d, t_d)
        if ef != ef:
            torch.cuda.synchronize()
            ef = (y2.float() - y.float()).abs().max().item() / (y.float().abs().max().item() + 1e-9)
            ed = (dx2.float() - dx.float()).abs().max().item() / (dx.float().abs().max().item() + 1e-9)
        cells.append(f"{t_f:5.1f}/{t_d:5.1f}")
    lib.xpt_conv2d_stream_tune(1, 512, 3, 80)
    print(f"{name:5s} {cin:4d}->{cout:3d} {OH:3d}x{OW:3d} | halo {t_of:5.1f}/{t_od:5.1f} | " + " | ".join(cells) +
          f" | {mb:5.1f} MB, {mb / best_f / 8:.3f}/{mb / best_d / 8:.3f} of peak | rel diff {ef:.1e}/{ed:.1e}", flush=True)
