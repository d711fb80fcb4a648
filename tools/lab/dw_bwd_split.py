"""Lab: data-gradient part vs weight-gradient part of xpt_dwconv_multi_bwd (lab knobs xpt_dwconv_tune(-21 / -22)) at the model's
shapes, hot, through the C ABI."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from xpt_mde_2021_amd.hip import conv as _conv, lib as _lib  # noqa: E402
from xpt_mde_2021_amd.model.model_util.layer_ops import same_pad  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
_conv._apply_env_tuning()          # XPT_DW_TUNE=<codes>


def timeit(fn, n=20):
    for _ in range(2):
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
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (3 * n)


B = 8
print("shape: both | data part only | weight part only (us)")
for C, ks, H, W, S, input_of in ((32, [7, 7, 5], 64, 208, 2, [0, 0, 0]), (16, [5, 7, 7, 5], 32, 104, 1, [0, 1, 1, 1]),
                                 (88, [5, 3, 3, 5, 3], 8, 26, 1, [0, 0, 1, 1, 1]), (88, [5, 3], 8, 26, 1, [0, 1]),
                                 (44, [5, 3, 3, 5, 3], 16, 52, 1, [0, 0, 1, 1, 1]), (176, [5, 3, 3, 5, 3], 4, 13, 1, [0, 0, 1, 1, 1]),
                                 (22, [5, 7, 7, 5], 32, 104, 2, [0, 1, 1, 1]), (88, [5, 7, 7, 5], 16, 52, 2, [0, 1, 1, 1])):
    g = torch.Generator().manual_seed(1)
    mk = lambda *s: torch.randn(*s, generator=g).to(dev, torch.bfloat16).contiguous(memory_format=torch.channels_last)  # noqa: E731
    n, OH, OW = len(ks), (H + 1) // 2 if S == 2 else H, (W + 1) // 2 if S == 2 else W
    nu = max(input_of) + 1
    xin = [mk(B, C, H, W) for _ in range(nu)]
    dxs = [torch.empty_like(x) for x in xin]
    ws = [(torch.randn(C, 1, k, k, generator=g) * 0.2).to(dev) for k in ks]
    dys = [mk(B, C, OH, OW) for _ in ks]
    pads = [same_pad(H, k, 2) + same_pad(W, k, 2) if S == 2 else (k // 2,) * 4 for k in ks]
    pts, pls = [pd[0] for pd in pads], [pd[2] for pd in pads]
    parts = [torch.empty(lib.xpt_dwconv_bwd_weight_chunks(B, OH, OW, C, k, S) * C * k * k, device=dev) for k in ks]
    P, I, PU = ctypes.c_void_p * n, ctypes.c_int * n, ctypes.c_void_p * nu

    def bwd():
        rc = lib.xpt_dwconv_multi_bwd(PU(*[x.data_ptr() for x in xin]), PU(*[d.data_ptr() for d in dxs]), nu,
                                      P(*[d.data_ptr() for d in dys]), P(*[w.data_ptr() for w in ws]), P(*[q.data_ptr() for q in parts]),
                                      I(*ks), I(*pts), I(*pls), I(*input_of), n, B, H, W, C, S, OH, OW, 1, 1,
                                      torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc

    res = []
    for knob in (-20, -22, -21, -24, -25):
        lib.xpt_dwconv_tune(knob)
        res.append(timeit(bwd))
    lib.xpt_dwconv_tune(-20)
    print(f"C={C} ks={ks} {H}x{W} s{S}: {res[0]:6.1f} | {res[1]:6.1f} | {res[2]:6.1f} || all rows in flight: both {res[3]:6.1f}, weight only {res[4]:6.1f}", flush=True)
