"""Deep decoder convolutions (dp_up4 / dp_up3, batch 8): the split-K tile kernels (csrc/xpt_conv_splitk.hip) against the
kernels they replace (csrc/xpt_conv.hip), forward and data gradient, over slice counts.

    python tools/bench_splitk.py [batch]
"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from xpt_mde_2021_amd.hip import conv as xc, lib as _lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ONE = len(sys.argv) > 2 and sys.argv[2] == "one"          # PMC passes: dp_up4_conv1 only, old + automatic plan, few launches
lib = _lib.load()
LAYERS = [
    ("up4a", 1056, 256, 4, 13, True), ("up4b", 432, 256, 8, 26, False),
    ("up3a", 256, 128, 8, 26, True), ("up3b", 216, 128, 16, 52, False),
    ("up2a", 128, 64, 16, 52, True), ("up2b", 96, 64, 32, 104, False),
    ("pose6", 256, 256, 2, 7, False), ("up3bd", 128, 216, 16, 52, False),
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


print(f"batch {B}; us per call: old fwd / dgrad | split-K (auto) | forced slices 1 2 4 8 16: fwd/dgrad")
for name, cin, cout, H, W, ups in LAYERS:
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
    ws = torch.empty(16 * B * OH * OW * max(cout, cp), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    flops = 2.0 * B * OH * OW * cout * k * k * cin

    def old_f():
        _lib.check(lib.xpt_conv2d_fwd(x.data_ptr(), e["fwd"].data_ptr(), bias.data_ptr(), y.data_ptr(), B, H, W, cp, cp, cout, k, k, s,
                                      pt, pl, OH, OW, cout, int(ups), 0.1, torch.cuda.current_stream().cuda_stream), "fwd")

    def old_d():
        _lib.check(lib.xpt_conv2d_bwd_data(g.data_ptr(), e["bwd"].data_ptr(), dx.data_ptr(), B, OH, OW, e["Np"], cout, cp, k, k, s,
                                           pt, pl, H, W, cp, int(ups), torch.cuda.current_stream().cuda_stream), "dgrad")

    def new_f():
        _lib.check(lib.xpt_conv2d_fwd_splitk(x.data_ptr(), e["fwd"].data_ptr(), bias.data_ptr(), y2.data_ptr(), B, H, W, cp, cp, cout,
                                             k, k, pt, pl, OH, OW, cout, int(ups), 0.1, ws.data_ptr(), ws.numel(),
                                             torch.cuda.current_stream().cuda_stream), "fwd splitk")

    def new_d():
        _lib.check(lib.xpt_conv2d_bwd_data_splitk(g.data_ptr(), e["bwd"].data_ptr(), dx2.data_ptr(), B, OH, OW, e["Np"], cout, cp, k, k,
                                                  pt, pl, H, W, cp, int(ups), ws.data_ptr(), ws.numel(),
                                                  torch.cuda.current_stream().cuda_stream), "dgrad splitk")

    t_of, t_od = timeit(old_f), timeit(old_d)
    cells = []
    ef = ed = float('nan')
    for force in ((0,) if ONE else (0, 1, 2, 4, 8, 16)):
        lib.xpt_conv2d_splitk_tune(1, force, 1, 1 << 30)
        try:
            t_f = timeit(new_f)
        except _lib.XptHipError:
            t_f = float("nan")
        try:
            t_d = timeit(new_d)
        except _lib.XptHipError:
            t_d = float("nan")
        if force == 0 and t_f == t_f and t_d == t_d:
            torch.cuda.synchronize()
            ef = (y2.float() - y.float()).abs().max().item() / (y.float().abs().max().item() + 1e-9)
            ed = (dx2.float() - dx.float()).abs().max().item() / (dx.float().abs().max().item() + 1e-9)
        cells.append(f"{t_f:5.1f}/{t_d:5.1f}")
    tile_only = []
    for force in (() if ONE else (4, 8, 16)):                # the tile kernel alone (enable = 2: no finishing launch)
        lib.xpt_conv2d_splitk_tune(2, force, 1, 1 << 30)
        tile_only.append(f"{timeit(new_f):5.1f}/{timeit(new_d):5.1f}")
    w4 = []
    for force in (() if ONE else (0,)):                      # four waves per workgroup (the default is eight on the 128-channel tile)
        lib.xpt_conv2d_splitk_tune(4, force, 1, 1 << 30)
        try:
            w4.append(f"{timeit(new_f):5.1f}/{timeit(new_d):5.1f}")
        except _lib.XptHipError:
            w4.append("n/a")
    lib.xpt_conv2d_splitk_tune(8, 0, 1024, 8192)
    print(f"        4 waves per workgroup (auto slices): " + " | ".join(w4))
    print(f"        tile kernel alone, 4 / 8 / 16 slices: " + " | ".join(tile_only))
    if ONE:
        break
    print(f"{name:5s} {cin:4d}->{cout:3d} {OH:3d}x{OW:3d} | old {t_of:5.1f}/{t_od:5.1f} | " + " | ".join(cells) +
          f" | rel diff {ef:.1e}/{ed:.1e}", flush=True)
