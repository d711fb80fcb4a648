"""Lab: would 48 physical channels (8-channel groups) beat 44 (4-channel groups) on the 16 x 52 stack?  Hot times of the
depthwise multi-layer launches and the single pointwise forward at both widths (C ABI)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from xpt_mde_2021_amd.hip import lib as _lib  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()


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


B, H, W = 8, 16, 52
ks, input_of = [5, 3, 3, 5, 3], [0, 0, 1, 1, 1]
S = lambda: torch.cuda.current_stream().cuda_stream  # noqa: E731
for C in (44, 48, 88, 96):
    g = torch.Generator().manual_seed(1)
    mk = lambda *s: torch.randn(*s, generator=g).to(dev, torch.bfloat16).contiguous(memory_format=torch.channels_last)  # noqa: E731
    n, nu = len(ks), 2
    xin = [mk(B, C, H, W) for _ in range(nu)]
    dxs = [torch.empty_like(x) for x in xin]
    ws = [(torch.randn(C, 1, k, k, generator=g) * 0.2).to(dev) for k in ks]
    ys = [mk(B, C, H, W) for _ in ks]
    pts = [k // 2 for k in ks]
    parts = [torch.empty(lib.xpt_dwconv_bwd_weight_chunks(B, H, W, C, k, 1) * C * k * k, device=dev) for k in ks]
    P, I, PU = ctypes.c_void_p * n, ctypes.c_int * n, ctypes.c_void_p * nu

    def fwd():
        assert lib.xpt_dwconv_multi_fwd(P(*[xin[u].data_ptr() for u in input_of]), P(*[w.data_ptr() for w in ws]),
                                        P(*[y.data_ptr() for y in ys]), I(*ks), I(*pts), I(*pts), n, B, H, W, C, 1, H, W, 1, 1, S()) == 0

    def bwd():
        assert lib.xpt_dwconv_multi_bwd(PU(*[x.data_ptr() for x in xin]), PU(*[d.data_ptr() for d in dxs]), nu,
                                        P(*[d.data_ptr() for d in ys]), P(*[w.data_ptr() for w in ws]), P(*[q.data_ptr() for q in parts]),
                                        I(*ks), I(*pts), I(*pts), I(*input_of), n, B, H, W, C, 1, H, W, 1, 1, S()) == 0

    M = B * H * W
    x = torch.randn(M, C, device=dev).bfloat16()
    w = torch.randn(C, C, device=dev).bfloat16()
    g_, b_, mu, var = (torch.rand(C, device=dev) + .5 for _ in range(4))
    ypre = torch.empty(M, C, device=dev, dtype=torch.bfloat16)
    y = torch.empty_like(ypre)

    def pw():
        assert lib.xpt_pwconv_bn_fwd(x.data_ptr(), w.data_ptr(), g_.data_ptr(), b_.data_ptr(), mu.data_ptr(), var.data_ptr(), 1e-3, None,
                                     ypre.data_ptr(), y.data_ptr(), M, C, C, C, S()) == 0

    print(f"C={C}: dw multi fwd {timeit(fwd):6.1f}  dw multi bwd {timeit(bwd):6.1f}  pw fwd {timeit(pw):6.1f}", flush=True)
