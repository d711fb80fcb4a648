"""Are the matrix-core convolution kernels bit-repeatable, launch after launch and under hipGraph replay?  (B = 8 shapes)"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from xpt_mde_2021_amd.hip import conv as xc, lib as _lib
lib = _lib.load()
B = 8
for name, cin, cout, k, s, H, W, ups in [("up0a", 32, 16, 3, 1, 64, 208, True), ("up0b", 17, 16, 3, 1, 128, 416, False),
                                         ("up1a", 64, 32, 3, 1, 32, 104, True), ("pose0", 15, 32, 5, 2, 128, 416, False)]:
    cp = xc.round_up(cin, 8)
    x = torch.randn(B, cp, H, W, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, k, k, device="cuda") / math.sqrt(cin * k * k)).contiguous(memory_format=torch.channels_last)
    Hl, Wl = H << ups, W << ups
    (pt, pb), (pl, pr) = xc.same_pad(Hl, k, s), xc.same_pad(Wl, k, s)
    OH, OW = -(-Hl // s), -(-Wl // s)
    g = torch.randn(B, cout, OH, OW, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    nsplit = lib.xpt_conv2d_bwd_weight_splits(B, cp, cout, k, k, s, OH, OW)
    n = cout * k * k * cin
    part = torch.full((nsplit * n,), float("nan"), dtype=torch.float32, device="cuda")
    def run():
        _lib.check(lib.xpt_conv2d_bwd_weight_partials(g.data_ptr(), x.data_ptr(), part.data_ptr(), part.numel(), B, H, W, cp, cin, cp,
                                                      cout, cout, k, k, s, pt, pl, OH, OW, int(ups),
                                                      torch.cuda.current_stream().cuda_stream), "wgrad")
    run(); torch.cuda.synchronize()
    first = part.clone()
    nan = int(torch.isnan(first).sum())
    dw = first.view(nsplit, -1).sum(0).view(cout, k, k, cin).permute(0, 3, 1, 2)
    xin = F.interpolate(x.float(), scale_factor=2, mode="nearest") if ups else x.float()
    xin = F.pad(xin, (pl, pr, pt, pb))
    ref = torch.ops.aten.convolution_backward(g.float(), xin, F.pad(w, (0, 0, 0, 0, 0, cp - cin)), None, [s, s], [0, 0], [1, 1], False, [0, 0], 1,
                                              [False, True, False])[1][:, :cin]
    err = float((dw - ref).abs().max() / ref.abs().max())
    diffs = []
    for it in range(5):
        part.fill_(float("nan")); run(); torch.cuda.synchronize()
        diffs.append(int((part != first).sum()) - 0)
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side): run()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr): run()
    gdiffs = []
    for it in range(4):
        part.fill_(float("nan")); gr.replay(); torch.cuda.synchronize()
        gdiffs.append(int((part != first).sum()))
    print(f"[repeat] {name}: splits {nsplit}, unwritten/NaN partial elements {nan}, rel err vs fp32 {err:.2e}, elements differing "
          f"on relaunch {diffs}, on graph replay {gdiffs}", flush=True)
